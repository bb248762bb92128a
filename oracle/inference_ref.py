"""Oracle: sliding-window inference and evaluation Dice.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Restates /root/reference/inference/inference3d.py:28-99,
inference/utils.py:29-43 and metric/utils.py:33-82 with stock torch ops on the CPU, parameterised by a forward
function (e.g. ``oracle.unet_ref.unet_forward`` bound to a state_dict).  Pinned by
tests/golden/infer_resunet_b8.npz (the real reference, ``make_golden_infer.py``).
"""
import torch
import torch.nn.functional as F


def split_idx(half_win, size, i):
    """inference/utils.py:29-43."""
    start = half_win * i
    end = start + half_win * 2
    if end > size:
        start, end = size - half_win * 2, size
    return start, end


def sliding_window(forward, img, window, classes):
    """inference3d.py:28-99."""
    B, C, D, H, W = img.shape
    wd, wh, ww = window
    origin = None
    if D < wd or H < wh or W < ww:
        origin = (D, H, W)
        img = F.pad(img, (0, max(0, ww - W), 0, max(0, wh - H), 0, max(0, wd - D)))
        B, C, D, H, W = img.shape
    hd, hh, hw = wd // 2, wh // 2, ww // 2
    out = torch.zeros((B, classes, D, H, W), dtype=img.dtype)
    cnt = torch.zeros((B, 1, D, H, W), dtype=img.dtype)
    for i in range(D // hd):
        for j in range(H // hh):
            for k in range(W // hw):
                d0, d1 = split_idx(hd, D, i)
                h0, h1 = split_idx(hh, H, j)
                w0, w1 = split_idx(hw, W, k)
                out[:, :, d0:d1, h0:h1, w0:w1] += F.softmax(forward(img[:, :, d0:d1, h0:h1, w0:w1]), dim=1)
                cnt[:, :, d0:d1, h0:h1, w0:w1] += 1
    out /= cnt
    return out if origin is None else out[:, :, :origin[0], :origin[1], :origin[2]]


def dice(pred, target, C):
    """calculate_dice (metric/utils.py:62-82); the returned summ includes the +1e-5."""
    pm = F.one_hot(pred.view(-1).long(), C).to(torch.float32)
    tm = F.one_hot(target.view(-1).long(), C).to(torch.float32)
    inter = (pm * tm).sum(0)
    summ = (pm + tm).sum(0)
    summ += 1e-5
    return 2 * inter / summ, inter, summ


def dice_split(pred, target, C, block_size=64 * 64 * 64):
    """calculate_dice_split (metric/utils.py:33-53)."""
    pred, target = pred.view(-1), target.view(-1)
    N = pred.shape[0]
    ts, ti = torch.zeros(C), torch.zeros(C)
    for s in range(0, N, block_size):
        _, i_, s_ = dice(pred[s:s + block_size], target[s:s + block_size], C)
        ti += i_
        ts += s_
    return 2 * ti / (ts + 1e-5), ti, ts
