"""Whole-image and sliding-window inference with the reference's call surface
(/root/reference/inference/inference3d.py:8-99).  The network forward runs on the HIP kernels; the per-window
softmax + accumulate + count and the final normalisation are fused kernels (csrc/inference_kernels.hip)."""
import ctypes as C

import torch
import torch.nn.functional as F

from .. import _lib
from ..ops import _dev_ok, _p, _stream
from .utils import split_idx


def _logits(net, x):
    pred = net(x)
    if isinstance(pred, (tuple, list)):     # deep-supervision nets return [out, aux_out]
        pred = pred[0]
    return pred.contiguous().float()


def _accumulate(logits, acc, counter, d0, h0, w0):
    _dev_ok(logits, acc, counter)
    B, K, wd, wh, ww = map(int, logits.shape)
    _, _, D, H, W = map(int, acc.shape)
    _lib.check(_lib.lib().cbim_softmax_accumulate(_p(logits), _p(acc), _p(counter), B, K, wd, wh, ww, D, H, W, d0, h0, w0,
                                                  _stream(logits)), "softmax_accumulate")


def _finalize(acc, counter, want_labels=False):
    B, K = int(acc.shape[0]), int(acc.shape[1])
    S = acc.numel() // (B * K)
    labels = torch.empty((B,) + tuple(acc.shape[2:]), dtype=torch.int64, device=acc.device) if want_labels else None
    _lib.check(_lib.lib().cbim_prob_finalize(_p(acc), _p(counter), _p(labels), B, K, S, _stream(acc)), "prob_finalize")
    return labels


def _label_gate():
    """The validation entry points are the engine's per-epoch hook for the deferred label check (functional.check_labels): the
    fused loss counts out-of-range labels on the device — also inside a replayed hipGraph — and this is where the count is read
    (one synchronisation per validated volume) and raised as the reference's CrossEntropyLoss / scatter_ would have at the
    offending training step."""
    from .. import functional as Fn
    if Fn._CHECK_LABELS not in ("", "0"):
        Fn.check_labels()


def inference_whole_image(net, img, args=None):
    """softmax(net(img), 1) — inference3d.py:8-26."""
    _label_gate()
    net.eval()
    with torch.no_grad():
        logits = _logits(net, img)
        acc = torch.zeros_like(logits)
        _accumulate(logits, acc, None, 0, 0, 0)
    return acc


def inference_sliding_window(net, img, args, return_labels=False):
    """Half-overlapping windows of args.window_size, probabilities averaged over the windows covering a voxel
    (inference3d.py:28-99).  With return_labels also the argmax map of validation.py:44 from the same pass."""
    _label_gate()
    net.eval()
    B, Cc, D, H, W = img.shape
    win_d, win_h, win_w = args.window_size
    flag = False
    if D < win_d or H < win_h or W < win_w:
        flag = True
        origin = (D, H, W)
        img = F.pad(img, (0, max(0, win_w - W), 0, max(0, win_h - H), 0, max(0, win_d - D)))
        B, Cc, D, H, W = img.shape
    hd, hh, hw = win_d // 2, win_h // 2, win_w // 2
    acc = torch.zeros((B, args.classes, D, H, W), dtype=torch.float32, device=img.device)
    counter = torch.zeros((B, 1, D, H, W), dtype=torch.float32, device=img.device)
    with torch.no_grad():
        for i in range(D // hd):
            for j in range(H // hh):
                for k in range(W // hw):
                    d0, d1 = split_idx(hd, D, i)
                    h0, h1 = split_idx(hh, H, j)
                    w0, w1 = split_idx(hw, W, k)
                    logits = _logits(net, img[:, :, d0:d1, h0:h1, w0:w1].contiguous())
                    _accumulate(logits, acc, counter, d0, h0, w0)
        labels = _finalize(acc, counter, return_labels)
    if flag:
        acc = acc[:, :, :origin[0], :origin[1], :origin[2]]
        if labels is not None:
            labels = labels[:, :origin[0], :origin[1], :origin[2]]
    return (acc, labels) if return_labels else acc
