"""Autograd glue: one ``torch.autograd.Function`` per fused block of the hot path.

Each Function's forward/backward is a hand-scheduled sequence of HIP kernel launches
(``ops``); autograd only chains the blocks, delivers parameter gradients to ``param.grad``
(so DDP / optimizer hooks fire per parameter as the backward proceeds) and sums the two
gradient contributions of a skip tensor.

Fusion map (reference: /root/reference/model/dim3/conv_layers.py):
  pre-activation ``conv(act(IN(x)))``      -> norm+act applied on the conv's input load,
                                              statistics of the output from the conv epilogue
  ``BasicBlock`` (conv_layers.py:86-94)     -> 2-3 conv launches; residual add in the epilogue;
                                              backward = wgrad + dgrad(+act' mask, +IN-backward
                                              sums in the epilogue) + one normalisation pass
"""
from __future__ import annotations

import os
import threading
from typing import NamedTuple, Optional

import weakref

import torch

from . import ops
from .ops import ACT, IN_EPS, ConvGeom

_COMPUTE_DTYPE: Optional[torch.dtype] = None
# labels outside [0, C): the reference raises (scatter_ / CrossEntropyLoss "Target out of bounds"); the kernel counts
# them in out[3].  Reading the count is a device synchronisation, so by default only the first calls of a process
# are checked (a wrong class count in a config shows up at once); CBIM_CHECK_LABELS=1 checks every call, =0 none.
_CHECK_LABELS = os.environ.get("CBIM_CHECK_LABELS", "first")
_label_checks_left = 4
_label_bad = {}          # device -> float32[1]: labels outside [0, C) seen by the loss calls that were not checked on the spot


def check_labels(reset: bool = True) -> int:
    """One device synchronisation for a whole epoch / validation pass: the number of out-of-range labels the loss has seen since
    the last call (the calls after the first few are not checked one by one — reading the count would stall the stream every
    step); raises IndexError like the reference's CrossEntropyLoss / scatter_ would have at the offending step."""
    global _label_checks_left
    bad = sum(int(t.item()) for t in _label_bad.values())
    if reset:
        for t in _label_bad.values():
            t.zero_()
        _label_checks_left = 4
    if bad:
        raise IndexError(f"cbim_amd: {bad} label(s) outside the class range since the last check (Target out of bounds)")
    return bad


def set_compute_dtype(dtype):
    global _COMPUTE_DTYPE
    if dtype in (None, "auto"):
        _COMPUTE_DTYPE = None
    elif dtype in ("bf16", "bfloat16", torch.bfloat16):
        _COMPUTE_DTYPE = torch.bfloat16
    elif dtype in ("fp32", "float32", torch.float32):
        _COMPUTE_DTYPE = torch.float32
    else:
        raise ValueError(f"unsupported compute dtype {dtype!r}")


def compute_dtype() -> torch.dtype:
    if _COMPUTE_DTYPE is not None:
        return _COMPUTE_DTYPE
    if torch.is_autocast_enabled():  # train.py --amp (fp16 autocast there; bf16 storage here)
        return torch.bfloat16
    return torch.float32


class FMap(NamedTuple):
    """Channels-last feature map + (lazily computed) InstanceNorm statistics."""
    t: torch.Tensor
    stats: Optional[torch.Tensor] = None


def ensure_stats(f: FMap, eps: float = IN_EPS) -> FMap:
    if f.stats is not None:
        return f
    with torch.no_grad():
        return FMap(f.t, ops.instnorm_stats(f.t.detach(), eps))


def _geom_c(x: torch.Tensor, cout: int, w: torch.Tensor, act: int) -> ConvGeom:
    """geometry of a convolution whose weight is `w` with `cout` output channels (Cout-concatenated pairs)"""
    k = tuple(int(i) for i in w.shape[2:])
    pad = tuple(i // 2 for i in k)
    return ConvGeom(x.dtype, int(x.shape[0]), tuple(x.shape[1:4]), int(w.shape[1]), int(cout), k, pad, act)


def _geom(x: torch.Tensor, w: torch.Tensor, act: int) -> ConvGeom:
    k = tuple(int(i) for i in w.shape[2:])
    pad = tuple(i // 2 for i in k)  # conv_layers.py:62,77
    return ConvGeom(x.dtype, int(x.shape[0]), tuple(x.shape[1:4]), int(w.shape[1]), int(w.shape[0]), k, pad, act)


def eager_only(fn):
    """Decorator of the model classes' forward: under the reference's optional `net = torch.compile(net)` wrapper
    (/root/reference/train.py:292-293) TorchDynamo must not trace into the engine — its operators are ctypes calls into
    libcbim_hip.so behind autograd Functions; the wrapped module then simply runs them eagerly (`net._orig_mod` is the engine
    module, train.py:106).
    The mark is dynamo's own attribute (`_torchdynamo_disable`, what torch.compiler.disable sets on the function it wraps),
    attached WITHOUT importing torch._dynamo: `import cbim_amd.model` then neither pays for that import (seconds) nor fails when
    it does; torch.compiler.disable itself is applied lazily, the first time the forward runs while dynamo is loaded."""
    import functools
    import sys
    if "torch._dynamo" in sys.modules:              # dynamo is already loaded at decoration time: its public API, nothing private
        dis = getattr(getattr(torch, "compiler", None), "disable", None)
        if dis is not None:
            try:
                return dis(fn)
            except Exception:
                pass
    state = {"wrapped": None}

    @functools.wraps(fn)
    def forward(*args, **kwargs):
        if "torch._dynamo" in sys.modules:          # someone (torch.compile) loaded dynamo: hand it the disabled function
            w = state["wrapped"]
            if w is None:
                dis = getattr(getattr(torch, "compiler", None), "disable", None)
                try:
                    w = dis(fn) if dis is not None else fn
                except Exception:                    # a broken dynamo install must not take eager training down with it
                    w = fn
                state["wrapped"] = w
            return w(*args, **kwargs)
        return fn(*args, **kwargs)

    forward._torchdynamo_disable = True             # dynamo skips frames of functions carrying this mark
    return forward


class _GradAwareFunction(torch.autograd.Function):
    """autograd.Function whose forward can tell whether the CALLER ran under torch.no_grad() (inside forward() grad mode
    is always off, and ctx.needs_input_grad only mirrors the inputs' requires_grad flags): sliding-window inference and
    validation must not pack the dgrad weight layouts they never use."""
    _tls = threading.local()             # per thread: a validating thread under no_grad must not flip a training thread's flag

    @classmethod
    def apply(cls, *args):
        _GradAwareFunction._tls.caller_grad = torch.is_grad_enabled()
        return super().apply(*args)


def _training(ctx) -> bool:
    return getattr(_GradAwareFunction._tls, "caller_grad", True) and any(ctx.needs_input_grad)


class StemFn(torch.autograd.Function):
    """inconv.conv1: raw Conv3d, NCDHW fp32 in -> channels-last out (unet_utils.py:14,19)."""

    @staticmethod
    def forward(ctx, x, w, out_dtype):
        x = x.contiguous().float()
        wd = w.detach().contiguous()
        pad = tuple(int(i) // 2 for i in w.shape[2:])
        y = ops.stem_fwd(x, wd, pad, out_dtype)
        ctx.save_for_backward(x)
        ctx.w_shape, ctx.pad, ctx.w_param = tuple(w.shape), pad, ops.slot_of(w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dw = ops.stem_wgrad(x, dy.contiguous(), ctx.w_shape, ctx.pad, out=ops.grad_slot(ctx.w_param))
        return None, dw, None


def _fusable(x: torch.Tensor, cout: int) -> bool:
    """conv1 + shortcut conv can run as one Cout-/K-concatenated GEMM when the split lands on a
    64-byte channel chunk (dgrad) and on a 32-channel block (wgrad)."""
    kc = 32 if x.dtype == torch.bfloat16 else 16
    return cout % kc == 0 and cout % 32 == 0


# Round 3 ("materialise act(IN(x)) once", profiles/r03_*; decided): every pre-activation conv input of a ReLU BasicBlock is written
# ONCE as a = act(IN(x)) by a streaming pass and the forward conv, the weight-gradient staging and the dgrad mask all read `a` as it
# is (pure LDS-DMA, no normalisation in LDS, no statistics tables).  Other activations normalise on load inside every consumer.


def fused_up_block(block) -> bool:
    """True when the decoder level's first block can take the fused up-sample + concat + activation path
    (UpBlockFirstFn): a ReLU BasicBlock with a shortcut conv."""
    from .model.dim3.conv_layers import BasicBlock, ConvNormAct
    return (isinstance(block, BasicBlock) and isinstance(block.shortcut, ConvNormAct)
            and block.conv1.act_code == ACT["relu"] and block.conv1.norm_kind == "in")


def _bb_fwd(ctx, xshape, xin, sin, ident, w1, w2, wsc, act, want_out_stats, mat, train):
    """conv1 (+ shortcut conv as one Cout-concatenated GEMM where the channel counts allow) and conv2 with the residual add
    of a pre-activation BasicBlock.  xin / sin: the block input as the convolutions read it (activated tensor + None when
    materialised, raw tensor + statistics otherwise); ident: the identity-shortcut tensor (None with a shortcut conv)."""
    cout = int(w1.shape[0])
    fused = wsc is not None and _fusable(xshape, cout)
    wdsc = None
    if fused:
        gc = _geom_c(xshape, 2 * cout, w1, act)
        wp, wd1 = ops.packed_weights((w1, wsc), gc, train)
        ycat, scat = ops.conv_fwd(xin, wp, gc, in_stats=sin, want_stats=True)
        y1, res = ycat[..., :cout], ycat[..., cout:]
        s1 = scat[:, :cout].contiguous()
        g1 = gsc = None
    else:
        gc = None
        g1 = _geom(xshape, w1, act)
        wp, wd1 = ops.packed_weights((w1,), g1, train)
        y1, s1 = ops.conv_fwd(xin, wp, g1, in_stats=sin, want_stats=True)
        if wsc is not None:
            gsc = _geom(xshape, wsc, act)
            wpsc, wdsc = ops.packed_weights((wsc,), gsc, train)
            res, _ = ops.conv_fwd(xin, wpsc, gsc, in_stats=sin)
        else:
            gsc = None
            res = ident
    g2 = _geom(y1, w2, act)
    wp2, wd2 = ops.packed_weights((w2,), g2, train)
    yin, s1in = (ops.norm_act_fwd(y1, s1, act), None) if mat else (y1, s1)
    out, so = ops.conv_fwd(yin, wp2, g2, in_stats=s1in, res=res, want_stats=want_out_stats)
    ctx.packed = (wd1, wd2, wdsc)
    ctx.w_params = (ops.slot_of(w1), ops.slot_of(w2), ops.slot_of(wsc))
    ctx.geoms = (g1, g2, gsc, gc)
    ctx.act, ctx.mat = act, mat
    if so is None:
        so = torch.empty(0, device=xin.device)
    ctx.mark_non_differentiable(so)
    ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics output
    return out, so, y1, s1, yin


def _bb_bwd(ctx, dout, cx, cxs, mxs, y1, s1, cy, cys, mys, w1):
    """backward of _bb_fwd down to the masked gradient of the block input: (gx, sums1, dw1, dw2, dwsc).
    cx / cy: the conv inputs as the forward saw them (+ statistics or None); mxs / mys: statistics for the dgrad mask (None:
    cx / cy are the activated tensors themselves and mask the gradient by [a > 0])."""
    g1, g2, gsc, gc = ctx.geoms
    wd1, wd2, wdsc = ctx.packed
    act = ctx.act
    p1, p2, psc = ctx.w_params
    dw2 = ops.conv_wgrad(cy, cys, dout, g2, out=ops.grad_slot(p2))
    gy1, sums2 = ops.conv_dgrad(dout, wd2, g2, mask_x=cy, mask_stats=mys)
    dy1 = ops.norm_bwd_apply(gy1, y1, s1, sums2, act, masked=False)
    if gc is not None:
        # conv1 + shortcut as one GEMM: dy = [dy1 | dout]
        cout = int(w1.shape[0])
        pair = ops.grad_slot_pair(p1, psc)
        dwcat = ops.conv_wgrad(cx, cxs, dy1, gc, dy2=dout, out=None if pair is None else pair[0])
        dw1, dwsc = (dwcat[:cout], dwcat[cout:]) if pair is None else pair[1:]
        gx, sums1 = ops.conv_dgrad(dy1, wd1, gc, mask_x=cx, mask_stats=mxs, dy2=dout)
        return gx, sums1, dw1, dw2, dwsc
    # conv1 (+ shortcut conv share act(IN(x)))
    dw1 = ops.conv_wgrad(cx, cxs, dy1, g1, out=ops.grad_slot(p1))
    if gsc is not None:
        dwsc = ops.conv_wgrad(cx, cxs, dout, gsc, out=ops.grad_slot(psc))
        gx_u, _ = ops.conv_dgrad(dy1, wd1, g1)
        gx, sums1 = ops.conv_dgrad(dout, wdsc, gsc, mask_x=cx, mask_stats=mxs, accumulate=gx_u)
    else:
        dwsc = None
        gx, sums1 = ops.conv_dgrad(dy1, wd1, g1, mask_x=cx, mask_stats=mxs)
    return gx, sums1, dw1, dw2, dwsc


class BasicBlockFn(_GradAwareFunction):
    """BasicBlock.forward (conv_layers.py:86-94) with pre-activation ConvNormAct (:48-49).

    inputs : x (raw, pre-norm), its statistics, w1, w2, wsc (or None), act code
    outputs: out = conv2(a(conv1(a(x)))) + shortcut, statistics of out (non-differentiable)

    conv1 and the shortcut conv read the same act(IN(x)) (InstanceNorm is parameter-free), so when
    the channel counts allow they run as ONE convolution with Cout-concatenated weights (forward,
    wgrad) / K-concatenated inputs (dgrad): the halo is staged and normalised once.

    ReLU blocks (every shipped configuration) materialise a = relu(IN(.)) of both conv inputs once
    : zero padding after the activation (conv_layers.py:48-49) is then simply the
    zero halo of a raw convolution.
    """

    @staticmethod
    def forward(ctx, x, x_stats, w1, w2, wsc, act, want_out_stats):
        train = _training(ctx)
        mat = act == ACT["relu"]
        xin, sin = (ops.norm_act_fwd(x, x_stats, act), None) if mat else (x, x_stats)
        out, so, y1, s1, yin = _bb_fwd(ctx, x, xin, sin, x, w1, w2, wsc, act, want_out_stats, mat, train)
        none = torch.empty(0)
        ctx.save_for_backward(x, x_stats, y1, s1, w1, w2, wsc if wsc is not None else none,
                              xin if mat and train else none, yin if mat and train else none)
        return out, so

    @staticmethod
    def backward(ctx, dout, _dso):
        x, x_stats, y1, s1, w1, w2, wsc, ax, ay1 = ctx.saved_tensors
        act = ctx.act
        dout = dout.contiguous()
        if ctx.mat:
            # the conv inputs as the forward saw them: activated tensors, no statistics; the dgrad mask is [a > 0]
            cx, cxs, cy, cys, mxs, mys = ax, None, ay1, None, None, None
        else:
            cx, cxs, cy, cys, mxs, mys = x, x_stats, y1, s1, x_stats, s1
        gx, sums1, dw1, dw2, dwsc = _bb_bwd(ctx, dout, cx, cxs, mxs, y1, s1, cy, cys, mys, w1)
        identity = ctx.geoms[2] is None and ctx.geoms[3] is None
        dx = ops.norm_bwd_apply(gx, x, x_stats, sums1, act, masked=False, add=dout if identity else None)
        return dx, None, dw1, dw2, dwsc, None, None


class UpBlockFirstFn(_GradAwareFunction):
    """up_block's trilinear up-sampling + concatenation (unet_utils.py:69-71) fused with the decoder level's first
    BasicBlock (in_ch + out_ch -> out_ch, always with a shortcut conv): the block reads a = relu(IN([skip | up(low)])),
    which ONE pass writes straight from `low` and `skip`; the raw concatenation is never stored.  Backward: the
    InstanceNorm backward of the (re-formed) concatenation writes dskip and the fine-resolution gradient of the
    up-sampled part, which the transposed trilinear gather turns into dlow."""

    @staticmethod
    def forward(ctx, low, skip, skip_stats, w1, w2, wsc, act, want_out_stats, skip_first):
        train = _training(ctx)
        up_st = ops.up_stats(low, skip.shape[1:4])
        stats_cat = torch.cat([skip_stats, up_st] if skip_first else [up_st, skip_stats], 1)
        a = ops.upcat_act_fwd(low, skip, stats_cat, act, skip_first)
        out, so, y1, s1, yin = _bb_fwd(ctx, a, a, None, None, w1, w2, wsc, act, want_out_stats, True, train)
        none = torch.empty(0)
        ctx.save_for_backward(low, skip, stats_cat, y1, s1, w1, w2, wsc, a if train else none, yin if train else none)
        ctx.skip_first = skip_first
        return out, so

    @staticmethod
    def backward(ctx, dout, _dso):
        low, skip, stats_cat, y1, s1, w1, w2, wsc, a, ay1 = ctx.saved_tensors
        dout = dout.contiguous()
        gx, sums1, dw1, dw2, dwsc = _bb_bwd(ctx, dout, a, None, None, y1, s1, ay1, None, None, w1)
        dlow, dskip = ops.upcat_norm_bwd(gx, low, skip, stats_cat, sums1, ctx.skip_first)
        return dlow, dskip, None, dw1, dw2, dwsc, None, None, None


class SingleConvFn(_GradAwareFunction):
    """SingleConv = post-activation ConvNormAct: act(IN(conv(x))) (conv_layers.py:51,56-68)."""

    @staticmethod
    def forward(ctx, x, w, act, need_dx):
        g = _geom(x, w, act)
        wp, wpd = ops.packed_weights((w,), g, bool(need_dx) and _training(ctx))
        z, sz = ops.conv_fwd(x, wp, g, want_stats=True)
        ctx.wpd = wpd
        y = ops.norm_act_fwd(z, sz, act)
        ctx.save_for_backward(x, z, sz, w)
        ctx.geom, ctx.act, ctx.need_dx, ctx.w_param = g, act, need_dx, ops.slot_of(w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, sz, w = ctx.saved_tensors
        g, act = ctx.geom, ctx.act
        dy = dy.contiguous()
        sums = ops.norm_bwd_sums(dy, z, sz, act, masked=True)
        dz = ops.norm_bwd_apply(dy, z, sz, sums, act, masked=True)
        dw = ops.conv_wgrad(x, None, dz, g, out=ops.grad_slot(ctx.w_param))
        dx = None
        if ctx.need_dx:
            dx, _ = ops.conv_dgrad(dz, ctx.wpd, g)
        return dx, dw, None, None


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool3d(scale) (unet_utils.py:36)."""

    @staticmethod
    def forward(ctx, x, scale):
        y, idx = ops.maxpool_fwd(x, scale)
        ctx.save_for_backward(idx)
        ctx.in_shape, ctx.scale = tuple(x.shape), tuple(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool_bwd(dy.contiguous(), idx, ctx.in_shape, ctx.scale), None


class UpCatFn(torch.autograd.Function):
    """F.interpolate(low, size=skip size, trilinear, align_corners=True) + cat (unet_utils.py:69-71)."""

    @staticmethod
    def forward(ctx, low, skip, skip_first, want_stats=False):
        ctx.low_shape, ctx.Cs, ctx.skip_first = tuple(low.shape), int(skip.shape[-1]), skip_first
        if want_stats:   # statistics of the concatenated tensor from the same pass (non-differentiable output)
            out, st = ops.upcat_fwd_stats(low, skip, skip_first)
            ctx.mark_non_differentiable(st)
            ctx.set_materialize_grads(False)
            return out, st
        return ops.upcat_fwd(low, skip, skip_first)

    @staticmethod
    def backward(ctx, dout, _dst=None):
        dlow, dskip = ops.upcat_bwd(dout.contiguous(), ctx.low_shape, ctx.Cs, ctx.skip_first)
        return dlow, dskip, None, None


class HeadFn(torch.autograd.Function):
    """outc = nn.Conv3d(base, classes, 1) with bias (unet.py:47): channels-last in, NCDHW fp32 out."""

    @staticmethod
    def forward(ctx, x, w, b):
        w2d = w.detach().reshape(w.shape[0], w.shape[1]).contiguous()
        ctx.save_for_backward(x, w2d)
        ctx.w_shape, ctx.w_param, ctx.b_param = tuple(w.shape), ops.slot_of(w), ops.slot_of(b)
        return ops.head_fwd(x, w2d, b.detach().contiguous())

    @staticmethod
    def backward(ctx, dlogits):
        x, w2d = ctx.saved_tensors
        sw = ops.grad_slot(ctx.w_param)
        dx, dw, db = ops.head_bwd(x, w2d, dlogits.contiguous().float(), need_dx=ctx.needs_input_grad[0],
                                  out_w=None if sw is None else sw.view(w2d.shape), out_b=ops.grad_slot(ctx.b_param))
        return dx, (sw if sw is not None else dw.reshape(ctx.w_shape)), db


class DiceCEFn(torch.autograd.Function):
    """(CE, Dice, CE+Dice) of train.py:212 in one pass over the logits (training/losses.py:18-58)."""

    @staticmethod
    def forward(ctx, logits, labels, weight):
        logits = logits.contiguous().float()
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        out, coef = ops.dice_ce_fwd(logits, labels, weight)
        # out[3] = number of labels outside [0, C): the reference raises (scatter_ / CrossEntropyLoss); reading the
        # count is a device synchronisation, so it is only checked on request
        global _label_checks_left
        capturing = logits.is_cuda and torch.cuda.is_current_stream_capturing()
        want = _CHECK_LABELS == "1" or (_CHECK_LABELS not in ("", "0") and _label_checks_left > 0)
        acc = _label_bad.get(logits.device)
        if acc is None and not capturing and _CHECK_LABELS not in ("", "0"):
            # the device-side count of out-of-range labels (functional.check_labels), allocated on the FIRST eager call so that it
            # exists — outside any graph's private pool — before a training step is captured into a hipGraph
            acc = _label_bad[logits.device] = torch.zeros(1, dtype=torch.float32, device=logits.device)
        if want and not capturing:
            _label_checks_left -= 1
            bad = int(out[3].item())
            if bad:
                raise IndexError(f"cbim_amd: {bad} label(s) outside [0, {int(logits.shape[1])}) (Target out of bounds)")
        elif acc is not None:
            # unchecked call: keep the count on the device.  Under hipGraph capture the add is captured with the step, so every
            # REPLAY keeps counting (the benchmarked mode); training.losses.check_labels() / the validation entry points read it
            acc += out[3:4].detach()
        ctx.save_for_backward(logits, labels, coef, weight if weight is not None else torch.empty(0))
        ctx.has_w = weight is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        logits, labels, coef, weight = ctx.saved_tensors
        g2 = torch.stack([gout[0] + gout[2], gout[1] + gout[2]]).float().contiguous()
        dz = ops.dice_ce_bwd(logits, labels, weight if ctx.has_w else None, coef, g2)
        return dz, None, None


class DicePerClassFn(torch.autograd.Function):
    """DiceLoss(reduce=False): the vector of per-class terms 1 - dice_c (training/losses.py:48-50).  The mean Dice loss is linear
    in the per-class coefficients the forward kernel leaves for the backward (coef[c] = -d dice_c / dTP_c / C, ...), so an
    upstream gradient g_c per class is the same backward kernel with coefficients scaled by C g_c."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous().float()
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        out, coef = ops.dice_ce_fwd(logits, labels, None)
        Cc = int(logits.shape[1])
        ctx.save_for_backward(logits, labels, coef)
        return coef[2 * Cc + 1:3 * Cc + 1].clone()

    @staticmethod
    def backward(ctx, g):
        logits, labels, coef = ctx.saved_tensors
        Cc = int(logits.shape[1])
        scaled = coef.clone()
        gc = g.float() * Cc
        scaled[:Cc] *= gc
        scaled[Cc:2 * Cc] *= gc
        g2 = torch.zeros(2, dtype=torch.float32, device=logits.device)
        g2[1] = 1.0                                   # (device-side fill: capturable)
        return ops.dice_ce_bwd(logits, labels, None, scaled, g2), None


# ------------------------------------------------------------------------------------------------
# MedFormer blocks (reference: /root/reference/model/dim3/medformer_utils.py, conv_layers.py:126-238)
# ------------------------------------------------------------------------------------------------

def restat(stats: torch.Tensor, eps_from: float, eps_to: float) -> torch.Tensor:
    """(mean, rstd) computed with one epsilon -> the same moments under another.  The reference's
    BidirectionAttentionBlock.norm1 / PatchMerging.norm use the InstanceNorm3d default 1e-5 while every
    ConvNormAct norm is built with 1e-4 (medformer_utils.py:112,158 vs conv_layers.py:40)."""
    return ops.stats_restat(stats.contiguous(), eps_from, eps_to)


class NormConvFn(_GradAwareFunction):
    """y = conv(act(IN(x))) [+ res]  (pre-activation ConvNormAct, conv_layers.py:48-49); stats=None -> raw
    conv.  `se` (float [N,Cin], optional; act must be none) folds the SEBlock gate of conv_layers.py:159-175
    into the normalisation: IN(x*s) = (x-mean) * s*rsqrt(var*s^2+eps), so the gated tensor is never written.
    Returns (y, InstanceNorm statistics of y [eps 1e-4] or an empty tensor)."""

    @staticmethod
    def forward(ctx, x, stats, w, act, res, want_stats, se, eps_out):
        g = _geom(x, w, act)
        wp, wpd = ops.packed_weights((w,), g, bool(ctx.needs_input_grad[0]) and _training(ctx))
        st = stats
        if se is not None:
            assert act == 0 and stats is not None
            st, ctx.rz2 = ops.se_fold_fwd(stats.contiguous(), se.detach().float().contiguous(), IN_EPS)
        # round 6: a layer k_conv3_rw48 takes (SwinUNETR's 48-channel monai blocks) reads a = act(IN(x)) written ONCE by a
        # streaming pass — forward, weight gradient and the dgrad's mask all use `a` as it is (pure LDS-DMA operands)
        ctx.mat = (stats is not None and se is None and x.dtype == torch.bfloat16 and act in (ACT["relu"], ACT["lrelu"])
                   and ops.rw48_takes(g))
        a = None
        if ctx.mat:
            a = ops.norm_act_fwd(x, st, act)
            y, so = ops.conv_fwd(a, wp, g, res=res, want_stats=want_stats, eps=eps_out)
        else:
            y, so = ops.conv_fwd(x, wp, g, in_stats=st, res=res, want_stats=want_stats, eps=eps_out)
        ctx.save_for_backward(x, st if st is not None else torch.empty(0), se if se is not None else torch.empty(0),
                              a if (a is not None and _training(ctx)) else torch.empty(0))
        ctx.geom, ctx.act, ctx.wpd, ctx.has = g, act, wpd, (stats is not None, res is not None, se is not None)
        ctx.w_param = ops.slot_of(w)
        if so is None:
            so = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(so)
        ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics output
        return y, so

    @staticmethod
    def backward(ctx, dy, _dso):
        x, st, se, a = ctx.saved_tensors
        has_stats, has_res, has_se = ctx.has
        g, act = ctx.geom, ctx.act
        dy = ops.as_rows(dy)                # (a channel-slice view from torch.cat's backward is read in place)
        if ctx.mat:
            dw = ops.conv_wgrad(a, None, dy, g, out=ops.grad_slot(ctx.w_param)) if ctx.needs_input_grad[2] else None
            dx = None
            if ctx.needs_input_grad[0]:
                gx, sums = ops.conv_dgrad(dy, ctx.wpd, g, mask_x=a, mask_stats=None)
                dx = ops.norm_bwd_apply(gx, x, st, sums, act, masked=False)
            return dx, None, dw, None, (dy if has_res else None), None, None, None
        dw = ops.conv_wgrad(x, st if has_stats else None, dy, g, out=ops.grad_slot(ctx.w_param)) if ctx.needs_input_grad[2] else None
        dx = ds = None
        if ctx.needs_input_grad[0]:
            if has_stats:
                gx, sums = ops.conv_dgrad(dy, ctx.wpd, g, mask_x=x, mask_stats=st)
                dx = ops.norm_bwd_apply(gx, x, st, sums, act, masked=False)
                if has_se:
                    S = g.in_dhw[0] * g.in_dhw[1] * g.in_dhw[2]
                    ds = ops.se_fold_bwd(sums.contiguous(), se.detach().float().contiguous(), ctx.rz2, IN_EPS, S)
            else:
                dx, _ = ops.conv_dgrad(dy, ctx.wpd, g)
        return dx, None, dw, None, (dy if has_res else None), None, ds, None


class DualRawConvFn(_GradAwareFunction):
    """Two bias-free convolutions of the SAME raw tensor — monai's UnetResBlock with a channel change reads x through conv1 (k^3)
    and through the 1x1x1 residual conv3 (/root/reference/model/dim3/swin_unetr.py:129-226) — as one autograd node: forward is the
    two launches (each with its output statistics), backward computes the second input gradient INTO the first (the kernels'
    accumulate operand) instead of leaving autograd an element-wise add over the block input (400 MB at 128^3 x 96 channels)."""

    @staticmethod
    def forward(ctx, x, w1, w3, eps):
        g1, g3 = _geom(x, w1, 0), _geom(x, w3, 0)
        train = bool(ctx.needs_input_grad[0]) and _training(ctx)
        wp1, wd1 = ops.packed_weights((w1,), g1, train)
        wp3, wd3 = ops.packed_weights((w3,), g3, train)
        z1, s1 = ops.conv_fwd(x, wp1, g1, want_stats=True, eps=eps)
        r, s3 = ops.conv_fwd(x, wp3, g3, want_stats=True, eps=eps)
        ctx.save_for_backward(x)
        ctx.cfg = (g1, g3, wd1, wd3)
        ctx.w_params = (ops.slot_of(w1), ops.slot_of(w3))
        ctx.mark_non_differentiable(s1, s3)
        ctx.set_materialize_grads(False)
        return z1, s1, r, s3

    @staticmethod
    def backward(ctx, dz1, _ds1, dr, _ds3):
        (x,) = ctx.saved_tensors
        g1, g3, wd1, wd3 = ctx.cfg
        dz1 = ops.as_rows(dz1) if dz1 is not None else None
        dr = ops.as_rows(dr) if dr is not None else None
        dw1 = ops.conv_wgrad(x, None, dz1, g1, out=ops.grad_slot(ctx.w_params[0])) if (dz1 is not None and ctx.needs_input_grad[1]) else None
        dw3 = ops.conv_wgrad(x, None, dr, g3, out=ops.grad_slot(ctx.w_params[1])) if (dr is not None and ctx.needs_input_grad[2]) else None
        dx = None
        if ctx.needs_input_grad[0]:
            if dz1 is not None:
                dx, _ = ops.conv_dgrad(dz1, wd1, g1)
            if dr is not None:
                dx, _ = ops.conv_dgrad(dr, wd3, g3, accumulate=dx)
        return dx, dw1, dw3, None


class DWConvFn(torch.autograd.Function):
    """Depthwise conv of act(IN(x)) (stats=None: of x) — DepthwiseSeparableConv.depthwise
    (conv_layers.py:137-145) / MBConv.depthwise (:211).  Also returns the per-(n,c) mean of the output
    (the SEBlock squeeze, conv_layers.py:163) as a differentiable output and the statistics (eps 1e-4)."""

    @staticmethod
    def forward(ctx, x, stats, w, act, want_mean):
        k = tuple(int(i) for i in w.shape[2:])
        w2d = w.detach().reshape(w.shape[0], -1).contiguous()
        y = ops.dwconv(x, w2d, k, in_stats=stats, act=act)
        ctx.save_for_backward(x, stats if stats is not None else torch.empty(0), w2d)
        ctx.k, ctx.act, ctx.has_stats, ctx.w_shape = k, act, stats is not None, tuple(w.shape)
        if want_mean:
            ys = ops.instnorm_stats(y, IN_EPS)
            mean = ys[..., 0].contiguous()
        else:
            ys = torch.empty(0, device=x.device)
            mean = torch.empty(0, device=x.device)
        ctx.want_mean = want_mean
        ctx.S = int(x.shape[1]) * int(x.shape[2]) * int(x.shape[3])
        ctx.mark_non_differentiable(ys)
        ctx.set_materialize_grads(False)
        return y, mean, ys

    @staticmethod
    def backward(ctx, dy, dmean, _dys):
        x, stats, w2d = ctx.saved_tensors
        st = stats if ctx.has_stats else None
        dy = dy.contiguous()
        bias = (dmean / ctx.S).float().contiguous() if (ctx.want_mean and dmean is not None) else None
        dw = ops.dwconv_wgrad(x, st, ctx.act, dy, ctx.k, dy_bias=bias).reshape(ctx.w_shape)
        dx = None
        if ctx.needs_input_grad[0]:
            gy = ops.dwconv(dy, w2d, ctx.k, bias=bias, flip=True)
            if st is not None:
                sums = ops.norm_bwd_sums(gy, x, st, ctx.act, masked=True)
                dx = ops.norm_bwd_apply(gy, x, st, sums, ctx.act, masked=True)
            else:
                dx = gy
        return dx, None, dw, None, None


class ChannelMeanFn(torch.autograd.Function):
    """The SEBlock squeeze (conv_layers.py:163,171) of a tensor whose InstanceNorm statistics are already known: mean = stats[..., 0]
    (no pass over x); backward hands x the per-channel constant dmean / S as a broadcast view, which autograd adds to x's other
    gradient in one pass.  (MBConv gets the same term through DWConvFn's gradient bias; this is FusedMBConv's, proj_type 'linear'.)"""

    @staticmethod
    def forward(ctx, x, stats):
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        return stats[..., 0].contiguous()

    @staticmethod
    def backward(ctx, dmean):
        N, D, H, W, Cc = ctx.shape
        g = (dmean.float() / float(D * H * W)).to(ctx.dtype)
        return g[:, None, None, None, :].expand(ctx.shape), None


class SpaceToDepthFn(torch.autograd.Function):
    """PatchMerging's 8 strided slices + channel concat (medformer_utils.py:163-171)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.in_shape, ctx.scale = tuple(x.shape), tuple(scale)
        return ops.space_to_depth(x, scale)

    @staticmethod
    def backward(ctx, dy):
        return ops.depth_to_space(dy.contiguous(), ctx.in_shape, ctx.scale), None


class BidirAttnFn(torch.autograd.Function):
    """BidirectionAttention core (medformer_utils.py:63-97): qv [N,D,H,W,2*inner] feature rows,
    mq / mv float32 [N,M,inner] -> (feat_out [N,D,H,W,inner], map_out float32 [N,M,inner])."""

    @staticmethod
    def forward(ctx, qv, mq, mv, heads, scale):
        mq = mq.detach().float().contiguous()
        mv = mv.detach().float().contiguous()
        ctx.heads, ctx.scale = heads, scale
        ctx.gemm = ops.awg_eligible(qv, mq, heads)
        if ctx.gemm:
            # round 6: one wide head (config/lits: d_head = the channel count) or a few heads with > 64 codes (config/acdc) — the four
            # products on the row-GEMM kernels, the two
            # softmaxes between them (csrc/attn_gemm_kernels.hip); P and C are kept for the backward (bf16 [L, M] each)
            fo, mo, P, Cs = ops.bidir_attn_gemm_fwd(qv, mq, mv, heads, scale)
            ctx.save_for_backward(qv, mq, mv, P, Cs, mo)
            return fo, mo
        fo, mo, cs = ops.bidir_attn_fwd(qv, mq, mv, heads, scale)
        ctx.save_for_backward(qv, mq, mv, cs, mo)
        return fo, mo

    @staticmethod
    def backward(ctx, dfo, dmo):
        if ctx.gemm:
            qv, mq, mv, P, Cs, mo = ctx.saved_tensors
            dqv, dmq, dmv = ops.bidir_attn_gemm_bwd(qv, mq, mv, P, Cs, mo, dfo.contiguous(), dmo.float().contiguous(), ctx.heads, ctx.scale)
            return dqv, dmq, dmv, None, None
        qv, mq, mv, cs, mo = ctx.saved_tensors
        dqv, dmq, dmv = ops.bidir_attn_bwd(qv, mq, mv, cs, mo, dfo.contiguous(), dmo.float().contiguous(),
                                           ctx.heads, ctx.scale)
        return dqv, dmq, dmv, None, None


class SEGateFn(torch.autograd.Function):
    """SEBlock.excitation (conv_layers.py:159-175) on the float32 [N, C] channel means: sigmoid(W2 relu(W1 m + b1) + b2) — two launches
    forward, three backward (ops.se_gate_fwd / se_gate_bwd) instead of ~13 ATen launches on vectors of a few hundred numbers."""

    @staticmethod
    def forward(ctx, mean, w1, b1, w2, b2):
        mean, w1, w2 = mean.float().contiguous(), w1.contiguous(), w2.contiguous()
        gate, z1 = ops.se_gate_fwd(mean, w1, b1, w2, b2)
        ctx.save_for_backward(mean, w1, w2, z1, gate)
        ctx.has_b = (b1 is not None, b2 is not None)
        return gate

    @staticmethod
    def backward(ctx, dgate):
        mean, w1, w2, z1, gate = ctx.saved_tensors
        dmean, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgate.float().contiguous(), gate, z1, mean, w1, w2,
                                                    need_dmean=ctx.needs_input_grad[0], need_bias=ctx.has_b)
        return dmean, dw1, db1, dw2, db2


class MapQVFn(torch.autograd.Function):
    """norm2 + map_qv of BidirectionAttentionBlock on the semantic map (medformer_utils.py:36,66,113,127): smap float32 [B, C, M],
    w [2 inner, C] -> (map_q, map_v) float32 [B, M, inner] — one launch forward (InstanceNorm over the positions on load), two
    backward (weight gradient; input gradient with the InstanceNorm backward as its epilogue)."""

    @staticmethod
    def forward(ctx, smap, w, eps):
        w = w.contiguous()
        mq, mv, mapp, rstd = ops.map_qv_fwd(smap.contiguous(), w, eps)
        ctx.save_for_backward(w, mapp, rstd)
        return mq, mv

    @staticmethod
    def backward(ctx, dmq, dmv):
        w, mapp, rstd = ctx.saved_tensors
        ds, dw = ops.map_qv_bwd(dmq.float().contiguous(), dmv.float().contiguous(), w, mapp, rstd, need_dw=ctx.needs_input_grad[1])
        return (ds if ctx.needs_input_grad[0] else None), dw, None


class MapOutFn(torch.autograd.Function):
    """map_out projection of BidirectionAttention + the block's `+ semantic_map` (medformer_utils.py:40,93,137): mo float32
    [B, M, inner], w [C, inner], smap [B, C, M] -> [B, C, M].  One launch forward, two backward."""

    @staticmethod
    def forward(ctx, mo, w, smap):
        w, mo = w.contiguous(), mo.contiguous()
        ctx.save_for_backward(w, mo)
        return ops.map_out_fwd(mo, w, smap.contiguous())

    @staticmethod
    def backward(ctx, g):
        w, mo = ctx.saved_tensors
        g = g.float().contiguous()
        dmo, dw = ops.map_out_bwd(g, w, mo, need_dw=ctx.needs_input_grad[1])
        return dmo, dw, (g if ctx.needs_input_grad[2] else None)


class MapPoolFn(torch.autograd.Function):
    """SemanticMapGeneration tail (medformer_utils.py:218-228): fw = [feat | weight logits] rows ->
    float32 [N, Cf, M]."""

    @staticmethod
    def forward(ctx, fw, Cf):
        mp, cs = ops.colsoftmax_pool_fwd(fw, Cf)
        ctx.save_for_backward(fw, mp, cs)
        ctx.Cf = Cf
        return mp

    @staticmethod
    def backward(ctx, dmap):
        fw, mp, cs = ctx.saved_tensors
        return ops.colsoftmax_pool_bwd(fw, ctx.Cf, mp, cs, dmap.float().contiguous()), None


class TrilinearPlanesFn(torch.autograd.Function):
    """F.interpolate(aux_out, size, 'trilinear', align_corners=True) on NCDHW float32 (medformer.py:91)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_shape = tuple(x.shape)
        return ops.trilinear_planes_fwd(x.contiguous().float(), size)

    @staticmethod
    def backward(ctx, dy):
        return ops.trilinear_planes_bwd(dy.contiguous().float(), ctx.in_shape), None


# ------------------------------------------------------------------------------------------------
# SwinUNETR blocks (reference: /root/reference/model/dim3/swin_unetr.py; monai 1.1.0 conv blocks)
# ------------------------------------------------------------------------------------------------

class WindowAttnFn(torch.autograd.Function):
    """pad + roll + window_partition + WindowAttention core + window_reverse + roll back + crop
    (swin_unetr.py:467-490, 554-606) as one kernel.  qkv [B,D,H,W,3C]; returns [B,D,H,W,C].
    The gradient returned for `qkv_bias` is only the part that flows through window-PADDING tokens (their
    q/k/v are the bias itself); the part through real tokens reaches the bias via the qkv projection."""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, heads, window, shift, table_window):
        qkv = qkv.contiguous()
        bias = qkv_bias.detach().float().contiguous() if qkv_bias is not None else None
        tbl = table.detach().float().contiguous()
        out, lse = ops.window_attn_fwd(qkv, bias, tbl, heads, window, shift, table_window)
        ctx.save_for_backward(qkv, bias if bias is not None else torch.empty(0), tbl, out, lse)
        ctx.cfg = (heads, tuple(window), tuple(shift), tuple(table_window), bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias, tbl, out, lse = ctx.saved_tensors
        heads, window, shift, tw, has_bias = ctx.cfg
        dqkv, dtable, dbias = ops.window_attn_bwd(qkv, bias if has_bias else None, tbl, out, dout.contiguous(), lse, heads,
                                                  window, shift, tw)
        return dqkv, (dbias if has_bias else None), dtable, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm / F.layer_norm over the channel axis of channels-last token rows (swin_unetr.py:539,550,679,970-983):
    x float32 [..., C] (the residual stream), optional affine parameters; y in `out_dtype` (bf16 for the token Linears
    of the bf16 engine mode: the cast rides on the store)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        x = x.contiguous()
        w = weight.detach().float().contiguous() if weight is not None else None
        b = bias.detach().float().contiguous() if bias is not None else None
        y, rs = ops.layernorm_fwd(x, w, b, eps, out_dtype)
        ctx.save_for_backward(x, w if w is not None else torch.empty(0), rs)
        ctx.has_w, ctx.has_b = w is not None, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, rs = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        # weight and bias are independent (nn.LayerNorm(bias=False), or a bias-only affine): the column sums are computed
        # when either is present, each gradient is returned only for the parameter that was given
        dx, dg, db = ops.layernorm_bwd(dy, x, w if ctx.has_w else None, rs, ctx.has_w or ctx.has_b)
        return dx, (dg if ctx.has_w else None), (db if ctx.has_b else None), None, None


def layer_norm(x, weight, bias, eps, out_dtype=torch.float32):
    return LayerNormFn.apply(x, weight, bias, eps, out_dtype)


class LayerNormResFn(torch.autograd.Function):
    """(LN(x), x): the pre-norm residual pattern x + f(LN(x)) of a transformer block (swin_unetr.py:539-552) as ONE autograd node
    with two outputs — the second is x itself, handed on to the branch's last Linear as its residual operand.  Backward then
    receives the gradient of both paths at once and the LayerNorm backward kernel adds the residual one to its dx
    (cbim_layernorm_bwd's `add`): the separate fp32 add pass autograd would run over the stream is gone."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        x = x.contiguous()
        w = weight.detach().float().contiguous() if weight is not None else None
        b = bias.detach().float().contiguous() if bias is not None else None
        y, rs = ops.layernorm_fwd(x, w, b, eps, out_dtype)
        ctx.save_for_backward(x, w if w is not None else torch.empty(0), rs)
        ctx.has_w, ctx.has_b = w is not None, b is not None
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, gx):
        x, w, rs = ctx.saved_tensors
        if dy is None:                      # only the pass-through was used
            return gx, None, None, None, None
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        add = None
        if gx is not None:
            add = gx.contiguous()
            add = add if add.dtype == torch.float32 else add.float()
        dx, dg, db = ops.layernorm_bwd(dy, x, w if ctx.has_w else None, rs, ctx.has_w or ctx.has_b, add=add)
        return dx, (dg if ctx.has_w else None), (db if ctx.has_b else None), None, None


class TokenLinearFn(_GradAwareFunction):
    """nn.Linear over channels-last token rows on the engine's row-GEMM kernel (ops.token_linear, bf16 operands, fp32
    accumulation) with the neighbouring element-wise work fused — the SwinUNETR trunk's qkv / proj / MLP / patch-merging /
    patch-embedding Linears (/root/reference/model/dim3/swin_unetr.py:467-490,552,640-643,707-731):

        y = act_in(x) @ W^T + b  [+ res]          act_in: GELU when x is the MLP's stored pre-activation h (the activated
                                                  tensor is never written); res: the fp32 residual stream (y is then fp32)

    x [..., Cin] bf16 (LayerNorm output / attention output / h) or fp32 (patch tokens); W fp32 [Cout, Cin(, 1...)] master
    weights re-packed once per optimizer step (ops.PACKED).  Backward: dW = dy^T act_in(x) and db = column sums of dy on the
    engine's kernels (dy taken in fp32 or bf16 as it arrives: no cast pass), dx = (dy @ W) * act_in'(x) in bf16, and dy itself
    as the gradient of `res`."""

    @staticmethod
    def forward(ctx, x, w, b, act_in, res, out_dtype, need_dx, exact):
        Cout, Cin = int(w.shape[0]), int(w.numel() // w.shape[0])
        shape = tuple(x.shape[:-1])
        x2 = x.reshape(-1, Cin)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        g = ops.linear_geom(Cin, Cout)
        train = _training(ctx)
        wp, wpd = ops.packed_weights((w,), g, bool(need_dx) and train)
        # exact (fp32 rows only): also the residue image w - bf16(w) — the product keeps fp32 accuracy (the Linears the reference
        # evaluates in fp32 when only the blocks run under reduced precision: patch embedding, patch merging)
        exact = bool(exact) and x2.dtype == torch.float32
        wl, wld = ops.lo_weights(w, g, bool(need_dx) and train) if exact else (None, None)
        r2 = None
        if res is not None:
            r2 = res.reshape(-1, Cout)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        bias = b.detach().float().contiguous() if b is not None else None
        y = ops.token_linear(x2, wp, bias, Cout, act_in=act_in, res=r2, out_dtype=torch.float32 if res is not None else out_dtype,
                             w_lo=wl)
        ctx.save_for_backward(x2)
        ctx.cfg = (act_in, wpd, Cin, Cout, b is not None, res is not None, tuple(x.shape), tuple(w.shape), bool(need_dx), wld)
        ctx.w_param, ctx.b_param = ops.slot_of(w), ops.slot_of(b)
        return y.view(shape + (Cout,))

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        act_in, wpd, Cin, Cout, has_b, has_res, x_shape, w_shape, need_dx, wld = ctx.cfg
        d2 = dy.reshape(-1, Cout)
        if d2.dtype not in (torch.float32, torch.bfloat16):
            d2 = d2.float()
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        dw = db = dx = None
        if ctx.needs_input_grad[1]:
            sw = ops.grad_slot(ctx.w_param)
            dw = ops.token_linear_wgrad(x2, d2, act_in, out=None if sw is None else sw.view(Cout, Cin))
            dw = sw if sw is not None else dw.view(w_shape)
        if has_b and ctx.needs_input_grad[2]:
            cp = 8 if d2.dtype == torch.bfloat16 else 4
            db = ops.colsum(d2) if (Cout % cp == 0 and Cout // cp <= 256) else d2.float().sum(0)
        if need_dx and ctx.needs_input_grad[0]:
            # the forward layer's dgrad image is a [Cin x Cout] "weight": dx = dy @ W, times act_in'(x) where x is the stored
            # pre-activation (the MLP's second Linear)
            exact = wld is not None and d2.dtype == torch.float32
            dx = ops.token_linear(d2, wpd, None, Cin, mask=x2 if act_in else None, mask_act=act_in,
                                  out_dtype=torch.float32 if exact else torch.bfloat16, w_lo=wld if exact else None)
            dx = dx.view(x_shape)
        return dx, dw, db, None, (dy if has_res else None), None, None, None


def token_linear(x, weight, bias, act_in=0, res=None, out_dtype=torch.bfloat16, need_dx=True, exact=False):
    return TokenLinearFn.apply(x, weight, bias, act_in, res, out_dtype, need_dx, exact)


class ResNormFn(torch.autograd.Function):
    """Tail of monai's UnetResBlock: y = act(IN(a) + (IN(b) | b)) with the statistics of a (and b) given
    (they come out of the producing convolutions' epilogues)."""

    @staticmethod
    def forward(ctx, a, stats_a, b, stats_b, act):
        y = ops.resnorm_fwd(a, stats_a, b, stats_b, act)
        ctx.save_for_backward(a, stats_a, b, stats_b if stats_b is not None else torch.empty(0))
        ctx.act, ctx.has_b = act, stats_b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        a, sa, b, sb = ctx.saved_tensors
        da, db = ops.resnorm_bwd(ops.as_rows(dy), a, sa, b, sb if ctx.has_b else None, ctx.act,
                                 need_db=ctx.needs_input_grad[2])
        return da, None, db, None, None


class DepthToSpaceFn(torch.autograd.Function):
    """[N,D,H,W,8C] -> [N,2D,2H,2W,C]: the scatter half of ConvTranspose3d(k=2,s=2) (monai UnetrUpBlock.transp_conv)."""

    @staticmethod
    def forward(ctx, t, scale):
        N, D, H, W, Cm = map(int, t.shape)
        sD, sH, sW = scale
        ctx.scale = tuple(scale)
        return ops.depth_to_space(t, (N, D * sD, H * sH, W * sW, Cm // (sD * sH * sW)), scale)

    @staticmethod
    def backward(ctx, dy):
        return ops.space_to_depth(dy.contiguous(), ctx.scale), None


class UpCatSkipFn(torch.autograd.Function):
    """torch.cat((depth_to_space(t), skip), channel axis) of monai's UnetrUpBlock (/root/reference/model/dim3/swin_unetr.py:176-228:
    transp_conv -> cat -> conv_block) without the stored up-sampled tensor: the scatter half of ConvTranspose3d(k = s = 2) writes
    the first channels of the concatenated tensor directly, the skip tensor is copied behind it; the backward hands the skip's
    gradient out as a channel-slice view and gathers the other slice back into the GEMM layout."""

    @staticmethod
    def forward(ctx, t, skip, scale):
        N, D, H, W, Cs = map(int, skip.shape)
        sD, sH, sW = scale
        Cu = int(t.shape[-1]) // (sD * sH * sW)
        out = torch.empty((N, D, H, W, Cu + Cs), dtype=t.dtype, device=t.device)
        ops.depth_to_space_into(t.contiguous(), out, Cu, scale)
        out[..., Cu:].copy_(skip)
        ctx.cfg = (Cu, tuple(scale))
        return out

    @staticmethod
    def backward(ctx, g):
        Cu, scale = ctx.cfg
        g = g if g.is_contiguous() else g.contiguous()
        return ops.space_to_depth_from(g, Cu, scale), g[..., Cu:], None


class NormActFn(torch.autograd.Function):
    """y = act(IN(z)) with given statistics (post-activation ConvNormAct tail, conv_layers.py:51)."""

    @staticmethod
    def forward(ctx, z, stats, act):
        ctx.save_for_backward(z, stats)
        ctx.act = act
        return ops.norm_act_fwd(z, stats, act)

    @staticmethod
    def backward(ctx, dy):
        z, stats = ctx.saved_tensors
        dy = dy.contiguous()
        sums = ops.norm_bwd_sums(dy, z, stats, ctx.act, masked=True)
        return ops.norm_bwd_apply(dy, z, stats, sums, ctx.act, masked=True), None, None


class BiasAddFn(torch.autograd.Function):
    """x + bias over the channel axis of a channels-last tensor (a Conv3d / ConvTranspose3d bias applied after the GEMM:
    /root/reference/model/dim3/vnet.py:40,80,100).  The bias gradient is the fixed-order column sum of the engine
    (`k_colsum_partial` + finish), not an ATen reduction."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias.detach().to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        db = ops.colsum(dy.contiguous().reshape(-1, int(dy.shape[-1]))) if ctx.needs_input_grad[1] else None
        return dy, db


class ActFn(torch.autograd.Function):
    """y = act(x) as one streaming pass (the norm + activation kernel with identity statistics) — VNet's ELU after a
    residual add (/root/reference/model/dim3/vnet.py:73,95,119)."""

    @staticmethod
    def forward(ctx, x, act):
        ident = torch.zeros((int(x.shape[0]), int(x.shape[-1]), 2), dtype=torch.float32, device=x.device)
        ident[..., 1] = 1.0
        ctx.save_for_backward(x, ident)
        ctx.act = act
        return ops.norm_act_fwd(x, ident, act)

    @staticmethod
    def backward(ctx, dy):
        x, ident = ctx.saved_tensors
        zero = torch.zeros_like(ident)
        return ops.norm_bwd_apply(dy.contiguous(), x, ident, zero, ctx.act, masked=True), None


class BatchNormActFn(torch.autograd.Function):
    """nn.BatchNorm3d / ContBatchNorm3d followed by an optional activation on the streaming norm kernels with a per-channel
    affine (round 5: cbim_norm_affine_*): batch statistics over (N, D, H, W) pooled from the per-(n, c) moments of
    `k_partial_sums`, z = gamma * xh + beta, y = act(z), running-statistics update as F.batch_norm does.
    `use_batch_stats`: True = training-mode statistics (ContBatchNorm3d ALWAYS, vnet.py:22-33: F.batch_norm(training=True);
    nn.BatchNorm3d in train()), False = the running statistics (nn.BatchNorm3d in eval()).
    Backward, with dz = dy act'(z): a = mean(dz), b = mean(dz xh) over the batch; d beta = sum dz, d gamma = sum dz xh,
    dx = gamma rstd (dz - a - xh b) under batch statistics, gamma rstd dz under running statistics.  The affine is NOT folded
    into the statistics (round 4 did: its backward divided by gamma): a zero or tiny gamma is exact."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act, use_batch_stats=True):
        N, C = int(x.shape[0]), int(x.shape[-1])
        S = 1
        for d in x.shape[1:-1]:
            S *= int(d)
        def f32(t):
            return None if t is None else (t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous())
        # round 6: ONE launch for the per-channel arithmetic (batch statistics pooled in float64, running-statistics update in place,
        # statistics repeated per image, affine pairs) — it was ~25 float64 ATen launches on [C] vectors per BatchNorm
        rm, rv = f32(running_mean), f32(running_var)
        copy_back = use_batch_stats and running_mean is not None and (rm.data_ptr() != running_mean.data_ptr() or rv.data_ptr() != running_var.data_ptr())
        st = ops.instnorm_stats(x, eps) if use_batch_stats else None        # [N, C, (mean, rstd)] per image
        stats, affine = ops.bn_finish_fwd(st, S, eps, momentum, rm, rv, use_batch_stats, f32(weight), f32(bias), N, C)
        if copy_back:                                 # running statistics kept in another dtype / layout
            with torch.no_grad():
                running_mean.copy_(rm)
                running_var.copy_(rv)
        ctx.save_for_backward(x, stats, affine)
        ctx.act, ctx.S, ctx.batch = act, S, bool(use_batch_stats)
        ctx.has = (weight is not None, bias is not None)
        return ops.norm_affine_act_fwd(x, stats, affine, act)

    @staticmethod
    def backward(ctx, dy):
        x, stats, affine = ctx.saved_tensors
        N, C = int(x.shape[0]), int(x.shape[-1])
        dy = dy.contiguous()
        sums = ops.norm_affine_bwd_sums(dy, x, stats, affine, ctx.act, masked=True)     # per image: mean(dz), mean(dz xh)
        dgamma, dbeta, sums_f = ops.bn_finish_bwd(sums, float(N * ctx.S), affine, ctx.batch, need_gamma=ctx.has[0], need_beta=ctx.has[1])
        dx = ops.norm_affine_bwd_apply(dy, x, stats, affine, sums_f, ctx.act, masked=True) if ctx.needs_input_grad[0] else None
        return dx, dgamma, dbeta, None, None, None, None, None, None


_SLICE_PACK = {}


def _packed_slice(w, kd, geom, need_dgrad, planes=None):
    """(weight slice [Cout, Cin, 1, kH, kW], packed forward image, packed dgrad image | None) of one kd plane of a 5x5x5 weight,
    re-packed only when the parameter moved (its version, or a hipGraph replay / whole-table re-pack: ops.PACKED.epoch) — ADVICE
    r04: the slices were re-packed on every forward AND backward (10 launches per convolution and step)."""
    if not isinstance(w, torch.nn.Parameter):      # a temporary (e.g. the zero-padded weight of VNet's output tail): pack on the spot
        wk = w.detach()[:, :, kd:kd + 1].contiguous()
        wp, wpd = (ops.pack_weights_both(wk, geom) if need_dgrad else (ops.pack_weights(wk, geom, 0), None))
        return wk, wp, wpd
    key = (w.data_ptr(), tuple(w.shape), kd, geom.dtype)
    tag = (w._version, ops.PACKED.epoch, ops.PACKED.stale)
    e = _SLICE_PACK.get(key)
    if e is not None and e[4]() is not w:          # the address was recycled by another tensor
        e = None
    # (stale: a hipGraph replay moved the weights unseen; capturing: the pack launch must be IN the graph — a replay re-packs from the
    #  weights its own captured optimizer step left)
    capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
    if e is None or e[0] != tag or ops.PACKED.stale or capturing or (need_dgrad and e[3] is None):
        # planes: the caller's lazily built contiguous [kD, Cout, Cin, kH, kW] copy of the weight (ONE permute per convolution and
        # step instead of one strided slice copy per plane — round 6: 125 launches of the VNet step)
        wk = planes()[kd].unsqueeze(2) if planes is not None else w.detach()[:, :, kd:kd + 1].contiguous()
        wp, wpd = (ops.pack_weights_both(wk, geom) if need_dgrad else (ops.pack_weights(wk, geom, 0), None))
        if e is None:                              # weakly keyed: entries die with their parameter
            weakref.finalize(w, _SLICE_PACK.pop, key, None)
        e = _SLICE_PACK[key] = (tag, wk, wp, wpd, weakref.ref(w))
    return e[1], e[2], e[3]


class ConvSlicesFn(torch.autograd.Function):
    """nn.Conv3d with a kernel of more than 64 taps (VNet's 5x5x5, vnet.py:40,60,126) as the sum over kd of (1, kH, kW)
    convolutions of D-shifted slices on the implicit-GEMM kernel: slice kd reads the input planes d + kd - pD and accumulates
    IN PLACE into the output planes they feed (k_conv_igemm's residual operand = its own output tensor).  Backward: the same
    decomposition for the input gradient (accumulated in place) and one weight-gradient launch per (image, kd)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        N, D, H, W, Cin = map(int, x.shape)
        Cout, _, kD, kH, kW = map(int, w.shape)
        pD, pH, pW = kD // 2, kH // 2, kW // 2
        wd = w.detach()
        y = torch.empty((N, D, H, W, Cout), dtype=x.dtype, device=x.device)
        plans = []
        box = []

        def planes():                                             # built at most once per call, only when a slice is re-packed
            if not box:
                box.append(wd.permute(2, 0, 1, 3, 4).contiguous())
            return box[0]
        order = [pD] + [k for k in range(kD) if k != pD]          # the centre slice first: it covers every output plane
        for kd in order:
            o = kd - pD
            d0, d1 = max(0, -o), min(D, D - o)
            if d1 <= d0:
                continue
            geom = ConvGeom(x.dtype, 1, (d1 - d0, H, W), Cin, Cout, (1, kH, kW), (0, pH, pW), 0)
            wk, wp, _ = _packed_slice(w, kd, geom, bool(ctx.needs_input_grad[0]), planes)
            for n in range(N):
                ys = y[n:n + 1, d0:d1]
                ops.conv_igemm(geom.fwd, x[n:n + 1, d0 + o:d1 + o], wp, tuple(ys.shape), res=None if kd == pD else ys, out=ys)
            plans.append((kd, o, d0, d1, geom, wk))
        if bias is not None:
            y += bias.detach().to(y.dtype)
        ctx.save_for_backward(x)
        ctx.plans, ctx.w_shape, ctx.has_bias, ctx.w_ref = plans, tuple(w.shape), bias is not None, w
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        N = int(x.shape[0])
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros(ctx.w_shape, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        for kd, o, d0, d1, geom, wk in ctx.plans:
            wpd = _packed_slice(ctx.w_ref, kd, geom, True)[2] if dx is not None else None
            for n in range(N):
                xs, dys = x[n:n + 1, d0 + o:d1 + o], dy[n:n + 1, d0:d1]
                if dx is not None:
                    dxs = dx[n:n + 1, d0 + o:d1 + o]
                    ops.conv_igemm(geom.bwd, dys, wpd, tuple(dxs.shape), res=dxs, out=dxs)
                if dw is not None:
                    dw[:, :, kd:kd + 1] += ops.conv_wgrad(xs, None, dys, geom)
        db = ops.colsum(dy.reshape(-1, int(dy.shape[-1]))) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


# ------------------------------------------------------------------------------------------------
# strided convolution (down_block(pool=False), /root/reference/model/dim3/unet_utils.py:36-39)
# ------------------------------------------------------------------------------------------------

_STRIDE_IDX = {}


def _stride_plan(k, stride):
    """Per axis: (phases, taps', pad', {(tap', phase): original tap}) of the space-to-depth form of a strided convolution.
    y[o] = sum_t w[t] x[s o + t - p], p = k // 2.  With x index i = s q + phase: for s = 2, k = 3: t = 0 -> (q = o - 1, phase 1),
    t = 1 -> (o, 0), t = 2 -> (o, 1): a 2-tap kernel over q in {o - 1, o} with one leading pad row; k = 1: (o, 0) only."""
    plan = []
    for kk, ss in zip(k, stride):
        if ss == 1:
            plan.append((1, kk, kk // 2, {(t, 0): t for t in range(kk)}))
        elif ss == 2 and kk == 3:
            plan.append((2, 2, 1, {(0, 1): 0, (1, 0): 1, (1, 1): 2}))
        elif ss == 2 and kk == 1:
            plan.append((2, 1, 0, {(0, 0): 0}))
        else:
            raise NotImplementedError(f"cbim_amd: strided convolution with kernel {kk} / stride {ss} along an axis is not built")
    return plan


def strided_conv(a, w, stride):
    """Conv3d(stride, padding = k // 2, bias = False) of a channels-last tensor used AS IT IS (the caller has normalised /
    activated it): space-to-depth turns the stride into channels — y = conv_{stride 1}(s2d(a), W') with the taps of w scattered over
    (phase, tap') and zeros elsewhere — so the dense stride-1 kernels, their input gradient and their weight gradient do the work;
    W' is built from the fp32 master weight by one differentiable gather.  Odd extents are zero-padded to even first (= the
    convolution's own zero padding).  Returns the channels-last output [N, ceil(D / sD), ..., Cout]."""
    import torch.nn.functional as F
    k = tuple(int(i) for i in w.shape[2:])
    stride = tuple(int(i) for i in stride)
    plan = _stride_plan(k, stride)
    N, D, H, W_, Cin = map(int, a.shape)
    pads = [(-D) % stride[0], (-H) % stride[1], (-W_) % stride[2]]
    if any(pads):
        a = F.pad(a, (0, 0, 0, pads[2], 0, pads[1], 0, pads[0]))
    out_dhw = tuple((e + p_) // s_ for e, p_, s_ in zip((D, H, W_), pads, stride))
    key = (k, stride, str(w.device))
    idx = _STRIDE_IDX.get(key)
    if idx is None:
        K = k[0] * k[1] * k[2]
        ph = [p[0] for p in plan]
        tp = [p[1] for p in plan]
        rows = []
        for pd in range(ph[0]):
            for p_h in range(ph[1]):
                for pw in range(ph[2]):                                   # phase order of k_space_to_depth: 4 pd + 2 ph + pw
                    row = []
                    for td in range(tp[0]):
                        for th in range(tp[1]):
                            for tw in range(tp[2]):
                                ko = [plan[0][3].get((td, pd)), plan[1][3].get((th, p_h)), plan[2][3].get((tw, pw))]
                                row.append(K if None in ko else (ko[0] * k[1] + ko[1]) * k[2] + ko[2])
                    rows.append(row)
        idx = _STRIDE_IDX[key] = torch.tensor(rows, dtype=torch.long, device=w.device)     # [phases, taps'], K = "no tap"
    Cout = int(w.shape[0])
    w_ext = torch.cat([w.reshape(Cout, Cin, -1), w.new_zeros(Cout, Cin, 1)], -1)          # [Co, Ci, K + 1]
    tp = [p[1] for p in plan]
    w2 = w_ext[:, :, idx].permute(0, 2, 1, 3).reshape(Cout, idx.shape[0] * Cin, tp[0], tp[1], tp[2]).contiguous()
    x2 = SpaceToDepthFn.apply(a.contiguous(), stride) if stride != (1, 1, 1) else a
    pad2 = tuple(p[2] for p in plan)
    g = ConvGeom(x2.dtype, N, tuple(int(i) for i in x2.shape[1:4]), int(x2.shape[-1]), Cout, tuple(tp), pad2, 0)
    y = RawConvFn.apply(x2, w2, g)
    # a leading pad row with a 2-tap kernel yields one output row too many at the end of a strided axis
    return y[:, :out_dhw[0], :out_dhw[1], :out_dhw[2]].contiguous()


class RawConvFn(_GradAwareFunction):
    """y = conv(x) with an explicit geometry (any padding the kernels take), no normalisation, residual or statistics — the
    stride-1 convolution under strided_conv."""

    @staticmethod
    def forward(ctx, x, w, geom):
        wp, wpd = ops.packed_weights((w,), geom, bool(ctx.needs_input_grad[0]) and _training(ctx))
        y, _ = ops.conv_fwd(x, wp, geom)
        ctx.save_for_backward(x)
        ctx.geom, ctx.wpd = geom, wpd
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dw = ops.conv_wgrad(x, None, dy, ctx.geom) if ctx.needs_input_grad[1] else None
        dx = ops.conv_dgrad(dy, ctx.wpd, ctx.geom)[0] if ctx.needs_input_grad[0] else None
        return dx, dw, None


class GateFn(torch.autograd.Function):
    """x * psi with one psi per voxel (AttentionBlock.forward, attention_unet_utils.py:35)."""

    @staticmethod
    def forward(ctx, x, psi):
        psi = psi.contiguous().float()
        ctx.save_for_backward(x, psi)
        return ops.gate_fwd(x, psi)

    @staticmethod
    def backward(ctx, dy):
        x, psi = ctx.saved_tensors
        return ops.gate_bwd(dy.contiguous(), x, psi)
