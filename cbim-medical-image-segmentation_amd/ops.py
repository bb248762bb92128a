"""Thin tensor-level wrappers over the C ABI (``include/cbim_hip.h``).

Tensors are torch tensors used purely as device memory: activations are channels-last
``[N, D, H, W, C]`` contiguous, dtype ``torch.bfloat16`` (fast mode) or ``torch.float32``
(parity mode); InstanceNorm statistics are ``float32 [N, C, 2]`` (mean, rstd).
Every function launches asynchronously on the current HIP stream.  There is no CPU or
PyTorch fallback: tensors must live where the loaded kernel library executes
(``cuda`` for libcbim_hip.so; ``cpu`` only for the test-suite's kernel executor).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import weakref

import torch

from . import _lib
from ._lib import ConvDesc, check

ACT = {"none": 0, None: 0, "relu": 1, "lrelu": 2, "gelu": 3, "swish": 4, "silu": 4, "elu": 5}
IN_EPS = 1e-4  # /root/reference/model/dim3/conv_layers.py:40,42

# bench.py sets this to a list to collect (kernel, algorithmic flops, start event, end event, shape)
# around every matrix-core launch (HIP events on the launch stream); None = no instrumentation.
PROFILE = None


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 0
    raise TypeError(f"cbim_amd: unsupported activation dtype {t.dtype}")


def _dev_ok(*ts):
    want = "cpu" if _lib.backend() == "emu" else "cuda"
    for t in ts:
        if t is None:
            continue
        if t.device.type != want:
            raise RuntimeError(
                f"cbim_amd: tensor on '{t.device.type}' but the loaded kernel library "
                f"({_lib.backend()}) executes on '{want}' — there is no fallback path")
        if not t.is_contiguous() and not _is_row_view(t):
            raise RuntimeError("cbim_amd: non-contiguous tensor passed to a kernel")


def _rows_ok(*ts):
    """_dev_ok for the token-row entry points, which take a row stride: [rows, C'] channel slices of a wider row tensor pass"""
    _dev_ok(*[t if (t is None or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]) else t.as_strided((1,), (1,)) for t in ts])


def _is_row_view(t: torch.Tensor) -> bool:
    """channels-last [N,D,H,W,C'] view that selects a channel range of a wider contiguous tensor:
    unit channel stride, rows `row_stride` elements apart, voxels dense."""
    if t.dim() != 5 or t.stride(-1) != 1:
        return False
    rs = t.stride(3)
    N, D, H, W, _ = t.shape
    return t.stride(2) == W * rs and t.stride(1) == H * W * rs and t.stride(0) == D * H * W * rs


def _rs(t: torch.Tensor) -> int:
    """row stride (elements between consecutive voxels)"""
    return int(t.stride(3)) if t.dim() == 5 else int(t.shape[-1])


_WARM = False


def _stream(t: torch.Tensor):
    global _WARM
    if t.device.type == "cuda":
        s = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        if not _WARM and not torch.cuda.is_current_stream_capturing():
            check(_lib.lib().cbim_runtime_warmup(s), "runtime_warmup")
            _WARM = True
        return s
    return None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _spatial(x):
    return int(x.shape[1]) * int(x.shape[2]) * int(x.shape[3])


# ------------------------------------------------------------------------------------------------
# InstanceNorm pieces
# ------------------------------------------------------------------------------------------------

def instnorm_stats(x: torch.Tensor, eps: float = IN_EPS) -> torch.Tensor:
    _dev_ok(x)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    L = _lib.lib()
    P = L.cbim_stats_parts(S, Cc)
    part = torch.empty((N, P, Cc, 3), dtype=torch.float32, device=x.device)
    stats = torch.empty((N, Cc, 2), dtype=torch.float32, device=x.device)
    check(L.cbim_instnorm_stats(_dt(x), _p(x), _rs(x), N, S, Cc, eps, _p(part), P, _p(stats), _stream(x)),
          "instnorm_stats")
    return stats


def stats_finalize(partials: torch.Tensor, count: float, eps: float, mode: int) -> torch.Tensor:
    N, P, Cc, _ = partials.shape
    out = torch.empty((N, Cc, 2), dtype=torch.float32, device=partials.device)
    check(_lib.lib().cbim_stats_finalize(_p(partials), N, P, Cc, float(count), eps, mode, _p(out),
                                         _stream(partials)), "stats_finalize")
    return out


def stats_restat(stats, eps_from: float, eps_to: float):
    """(mean, rstd) records computed with one epsilon -> the same moments under another (one launch, fp64 inside)."""
    _dev_ok(stats)
    out = torch.empty_like(stats)
    check(_lib.lib().cbim_stats_restat(_p(stats), float(eps_from), float(eps_to), _p(out), stats.numel() // 2, _stream(stats)),
          "stats_restat")
    return out


def se_fold_fwd(stats, se, eps: float):
    """SE gate folded into the normalisation -> (stats' float32 [N,C,2], rz2 float64 [N,C])."""
    _dev_ok(stats, se)
    out = torch.empty_like(stats)
    rz2 = torch.empty(tuple(se.shape), dtype=torch.float64, device=se.device)
    check(_lib.lib().cbim_se_fold_fwd(_p(stats), _p(se), float(eps), _p(out), _p(rz2), se.numel(), _stream(stats)), "se_fold_fwd")
    return out, rz2


def se_fold_bwd(sums, se, rz2, eps: float, S: int):
    _dev_ok(sums, se, rz2)
    ds = torch.empty(tuple(se.shape), dtype=torch.float32, device=se.device)
    check(_lib.lib().cbim_se_fold_bwd(_p(sums), _p(se), _p(rz2), float(eps), float(S), _p(ds), se.numel(), _stream(sums)), "se_fold_bwd")
    return ds


def norm_act_fwd(x, stats, act: int):
    _dev_ok(x, stats)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    y = torch.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    check(_lib.lib().cbim_norm_act_fwd(_dt(x), _p(x), _rs(x), _p(stats), _p(y), Cc, N, S, Cc, act, _stream(x)),
          "norm_act_fwd")
    return y


def norm_bwd_sums(g, x, stats, act: int, masked: bool):
    """(mean(g'), mean(g'*xh)) per (n, c), g' = g*act'(xh) when masked."""
    _dev_ok(g, x, stats)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    L = _lib.lib()
    P = L.cbim_stats_parts(S, Cc)
    part = torch.empty((N, P, Cc, 3), dtype=torch.float32, device=x.device)
    check(L.cbim_norm_bwd_reduce(_dt(x), _p(g), _rs(g), _p(x), _rs(x), _p(stats), N, S, Cc, act, int(masked), _p(part),
                                 P, _stream(x)), "norm_bwd_reduce")
    return stats_finalize(part, S, 0.0, 1)


def norm_bwd_apply(g, x, stats, sums, act: int, masked: bool, add=None):
    _dev_ok(g, x, stats, sums, add)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    dx = torch.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    check(_lib.lib().cbim_norm_bwd_apply(_dt(x), _p(g), _rs(g), _p(x), _rs(x), _p(stats), _p(sums), _p(add),
                                         _rs(add) if add is not None else 0, _p(dx),
                                         Cc, N, S, Cc, act, int(masked), _stream(x)), "norm_bwd_apply")
    return dx


def norm_affine_act_fwd(x, stats, affine, act: int):
    """y = act(gamma * (x - mean) * rstd + beta); affine float32 [C, 2] = (gamma, beta)."""
    _dev_ok(x, stats, affine)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    y = torch.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    check(_lib.lib().cbim_norm_affine_act_fwd(_dt(x), _p(x), _rs(x), _p(stats), _p(affine), _p(y), Cc, N, S, Cc, act, _stream(x)),
          "norm_affine_act_fwd")
    return y


def norm_affine_bwd_sums(g, x, stats, affine, act: int, masked: bool):
    """per (n, c): (mean(g'), mean(g' * xh)), g' = g * act'(gamma xh + beta) when masked."""
    _dev_ok(g, x, stats, affine)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    L = _lib.lib()
    P = L.cbim_stats_parts(S, Cc)
    part = torch.empty((N, P, Cc, 3), dtype=torch.float32, device=x.device)
    check(L.cbim_norm_affine_bwd_reduce(_dt(x), _p(g), _rs(g), _p(x), _rs(x), _p(stats), _p(affine), N, S, Cc, act, int(masked),
                                        _p(part), P, _stream(x)), "norm_affine_bwd_reduce")
    return stats_finalize(part, S, 0.0, 1)


def norm_affine_bwd_apply(g, x, stats, affine, sums, act: int, masked: bool):
    """dx = rstd * (gamma g' - m1 - xh m2), sums = (m1, m2) = gamma * (mean g', mean g' xh)."""
    _dev_ok(g, x, stats, affine, sums)
    N, Cc, S = int(x.shape[0]), int(x.shape[-1]), _spatial(x)
    dx = torch.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    check(_lib.lib().cbim_norm_affine_bwd_apply(_dt(x), _p(g), _rs(g), _p(x), _rs(x), _p(stats), _p(affine), _p(sums), _p(dx), Cc,
                                                N, S, Cc, act, int(masked), _stream(x)), "norm_affine_bwd_apply")
    return dx


# ------------------------------------------------------------------------------------------------
# pooling / upsample+concat
# ------------------------------------------------------------------------------------------------

def bn_finish_fwd(st, S: int, eps: float, momentum: float, running_mean, running_var, use_batch: bool, gamma, beta, N: int, C: int):
    """per-image (mean, rstd) [N, C, 2] -> (batch statistics repeated per image [N, C, 2], affine [C, 2]); updates the running
    statistics in place in training mode (include/cbim_hip.h cbim_bn_finish_fwd)"""
    dev = (st if st is not None else running_mean).device
    _dev_ok(st, running_mean, running_var, gamma, beta)
    stats = torch.empty((N, C, 2), dtype=torch.float32, device=dev)
    affine = torch.empty((C, 2), dtype=torch.float32, device=dev)
    ref = st if st is not None else running_mean
    check(_lib.lib().cbim_bn_finish_fwd(_p(st), N, C, float(S), float(eps), float(momentum), _p(running_mean), _p(running_var),
                                        int(bool(use_batch)), _p(gamma), _p(beta), _p(stats), _p(affine), _stream(ref)), "bn_finish_fwd")
    return stats, affine


def bn_finish_bwd(sums, cnt: float, affine, use_batch: bool, need_gamma: bool = True, need_beta: bool = True):
    """per-image means of g' and g' xh [N, C, 2] -> (d gamma | None, d beta | None, sums for norm_affine_bwd_apply [N, C, 2])"""
    _dev_ok(sums, affine)
    N, C = int(sums.shape[0]), int(sums.shape[1])
    dg = torch.empty((C,), dtype=torch.float32, device=sums.device) if need_gamma else None
    db = torch.empty((C,), dtype=torch.float32, device=sums.device) if need_beta else None
    out = torch.empty((N, C, 2), dtype=torch.float32, device=sums.device)
    check(_lib.lib().cbim_bn_finish_bwd(_p(sums), N, C, float(cnt), _p(affine), int(bool(use_batch)), _p(dg), _p(db), _p(out),
                                        _stream(sums)), "bn_finish_bwd")
    return dg, db, out


def maxpool_fwd(x, scale: Sequence[int]):
    _dev_ok(x)
    N, D, H, W, Cc = map(int, x.shape)
    sD, sH, sW = map(int, scale)
    y = torch.empty((N, D // sD, H // sH, W // sW, Cc), dtype=x.dtype, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
    check(_lib.lib().cbim_maxpool3d_fwd(_dt(x), _p(x), _p(y), _p(idx), N, D, H, W, Cc, sD, sH, sW, _stream(x)),
          "maxpool3d_fwd")
    return y, idx


def maxpool_bwd(dy, idx, in_shape, scale):
    _dev_ok(dy, idx)
    N, D, H, W, Cc = map(int, in_shape)
    sD, sH, sW = map(int, scale)
    dx = torch.empty(tuple(in_shape), dtype=dy.dtype, device=dy.device)
    check(_lib.lib().cbim_maxpool3d_bwd(_dt(dy), _p(dy), _p(idx), _p(dx), N, D, H, W, Cc, sD, sH, sW,
                                        _stream(dy)), "maxpool3d_bwd")
    return dx


def upcat_fwd(low, skip, skip_first: bool = True):
    _dev_ok(low, skip)
    N, Dl, Hl, Wl, Cl = map(int, low.shape)
    _, D, H, W, Cs = map(int, skip.shape)
    out = torch.empty((N, D, H, W, Cs + Cl), dtype=low.dtype, device=low.device)
    check(_lib.lib().cbim_upcat_fwd(_dt(low), _p(low), _p(skip), _p(out), N, Dl, Hl, Wl, Cl, D, H, W, Cs,
                                    int(skip_first), _stream(low)), "upcat_fwd")
    return out


def upcat_fwd_stats(low, skip, skip_first: bool = True, eps: float = IN_EPS):
    """upcat_fwd plus the InstanceNorm statistics of the concatenated tensor from the same pass."""
    _dev_ok(low, skip)
    N, Dl, Hl, Wl, Cl = map(int, low.shape)
    _, D, H, W, Cs = map(int, skip.shape)
    L = _lib.lib()
    Ct, S = Cs + Cl, D * H * W
    cpc = 8 if low.dtype == torch.bfloat16 else 4
    if Ct // cpc > 256 or S >= 2 ** 31:
        out = upcat_fwd(low, skip, skip_first)
        return out, instnorm_stats(out, eps)
    out = torch.empty((N, D, H, W, Ct), dtype=low.dtype, device=low.device)
    P = L.cbim_stats_parts(S, Ct)
    part = torch.empty((N, P, Ct, 3), dtype=torch.float32, device=low.device)
    stats = torch.empty((N, Ct, 2), dtype=torch.float32, device=low.device)
    check(L.cbim_upcat_fwd_stats(_dt(low), _p(low), _p(skip), _p(out), N, Dl, Hl, Wl, Cl, D, H, W, Cs, int(skip_first),
                                 eps, _p(part), P, _p(stats), _stream(low)), "upcat_fwd_stats")
    return out, stats


UP_TILES = True   # LDS-tiled up-path kernels (up_tile_kernels.hip); off: the gather kernels (A/B, tests)
_UNSUPPORTED = -2   # CBIM_EUNSUPPORTED


UP_GRAM = True   # statistics of the up-sampled tensor from the coarse grid (k_up_gram_stats)


def up_stats(low, out_dhw, eps: float = IN_EPS):
    """InstanceNorm statistics of trilinear(low -> out_dhw, align_corners=True) without writing it."""
    _dev_ok(low)
    N, Dl, Hl, Wl, Cl = map(int, low.shape)
    D, H, W = (int(i) for i in out_dhw)
    L = _lib.lib()
    if UP_GRAM and D >= Dl and H >= Hl and W >= Wl:
        P = L.cbim_up_gram_parts(Dl, Hl, Wl)
        part = torch.empty((N, P, Cl, 3), dtype=torch.float32, device=low.device)
        stats = torch.empty((N, Cl, 2), dtype=torch.float32, device=low.device)
        check(L.cbim_up_stats_gram(_dt(low), _p(low), N, Dl, Hl, Wl, Cl, D, H, W, eps, _p(part), P, _p(stats), _stream(low)), "up_stats_gram")
        return stats
    if UP_TILES:
        P = L.cbim_up_tile_parts(D, H, W)
        part = torch.empty((N, P, Cl, 3), dtype=torch.float32, device=low.device)
        stats = torch.empty((N, Cl, 2), dtype=torch.float32, device=low.device)
        rc = L.cbim_up_stats_tile(_dt(low), _p(low), N, Dl, Hl, Wl, Cl, D, H, W, eps, _p(part), P, _p(stats), _stream(low))
        if rc != _UNSUPPORTED:
            check(rc, "up_stats_tile")
            return stats
    P = L.cbim_stats_parts(D * H * W, Cl)
    part = torch.empty((N, P, Cl, 3), dtype=torch.float32, device=low.device)
    stats = torch.empty((N, Cl, 2), dtype=torch.float32, device=low.device)
    check(L.cbim_up_stats(_dt(low), _p(low), N, Dl, Hl, Wl, Cl, D, H, W, eps, _p(part), P, _p(stats), _stream(low)), "up_stats")
    return stats


def upcat_act_fwd(low, skip, stats_cat, act: int, skip_first: bool = True):
    """act(IN([skip | up(low)])) (channel order by skip_first) in one pass; stats_cat float [N, Cs+Cl, 2]."""
    _dev_ok(low, skip, stats_cat)
    N, Dl, Hl, Wl, Cl = map(int, low.shape)
    _, D, H, W, Cs = map(int, skip.shape)
    out = torch.empty((N, D, H, W, Cs + Cl), dtype=low.dtype, device=low.device)
    if UP_TILES:
        rc = _lib.lib().cbim_upcat_act_fwd_tile(_dt(low), _p(low), _p(skip), _p(stats_cat), _p(out), N, Dl, Hl, Wl, Cl, D, H, W, Cs,
                                                int(skip_first), act, _stream(low))
        if rc != _UNSUPPORTED:
            check(rc, "upcat_act_fwd_tile")
            return out
    check(_lib.lib().cbim_upcat_act_fwd(_dt(low), _p(low), _p(skip), _p(stats_cat), _p(out), N, Dl, Hl, Wl, Cl, D, H, W, Cs,
                                        int(skip_first), act, _stream(low)), "upcat_act_fwd")
    return out


UP_SEPARABLE = True   # dup -> dlow as three 1-D passes; off: the one-pass gather (A/B, tests)


def _lin_adjoint(src, src_row, c_off, dst, outer, F, L, inner, vec=0):
    check(_lib.lib().cbim_lin_adjoint_axis(_dt(src), vec, _p(src), src_row, c_off, _p(dst), outer, F, L, inner, _stream(src)),
          "lin_adjoint_axis")


def up_adjoint(src, c_off: int, Cl: int, low_dhw):
    """Transposed trilinear(align_corners=True) interpolation of channels [c_off, c_off+Cl) of the channels-last src
    [N, D, H, W, Ct] down to low_dhw, as one reduction per axis (W, then H, then D)."""
    N, D, H, W, Ct = map(int, src.shape)
    Dl, Hl, Wl = (int(i) for i in low_dhw)
    kw = dict(dtype=src.dtype, device=src.device)
    t1 = torch.empty((N, D, H, Wl, Cl), **kw)
    _lin_adjoint(src, Ct, c_off, t1, N * D * H, W, Wl, Cl)
    if H != Hl:
        t2 = torch.empty((N, D, Hl, Wl, Cl), **kw)
        _lin_adjoint(t1, Wl * Cl, 0, t2, N * D, H, Hl, Wl * Cl)
        t1 = t2
    if D != Dl:
        t2 = torch.empty((N, Dl, Hl, Wl, Cl), **kw)
        _lin_adjoint(t1, Hl * Wl * Cl, 0, t2, N, D, Dl, Hl * Wl * Cl)
        t1 = t2
    return t1


def upcat_norm_bwd(g, low, skip, stats_cat, sums, skip_first: bool = True):
    """InstanceNorm backward over the virtual concatenation -> (dlow, dskip)."""
    _dev_ok(g, low, skip, stats_cat, sums)
    N, Dl, Hl, Wl, Cl = map(int, low.shape)
    _, D, H, W, Cs = map(int, skip.shape)
    dskip = torch.empty_like(skip)
    dlow = None if UP_SEPARABLE else torch.empty_like(low)
    dup = torch.empty((N, D, H, W, Cl), dtype=low.dtype, device=low.device)
    rc = _UNSUPPORTED
    if UP_TILES:
        rc = _lib.lib().cbim_upcat_norm_bwd_tile(_dt(low), _p(g), _p(low), _p(skip), _p(stats_cat), _p(sums), _p(dskip), _p(dlow),
                                                 _p(dup), N, Dl, Hl, Wl, Cl, D, H, W, Cs, int(skip_first), _stream(low))
        if rc != _UNSUPPORTED:
            check(rc, "upcat_norm_bwd_tile")
    if rc == _UNSUPPORTED:
        check(_lib.lib().cbim_upcat_norm_bwd(_dt(low), _p(g), _p(low), _p(skip), _p(stats_cat), _p(sums), _p(dskip), _p(dlow), _p(dup),
                                             N, Dl, Hl, Wl, Cl, D, H, W, Cs, int(skip_first), _stream(low)), "upcat_norm_bwd")
    if dlow is None:
        dlow = up_adjoint(dup, 0, Cl, (Dl, Hl, Wl))
    return dlow, dskip


def upcat_bwd(dout, low_shape, Cs: int, skip_first: bool = True):
    _dev_ok(dout)
    N, Dl, Hl, Wl, Cl = map(int, low_shape)
    _, D, H, W, Ct = map(int, dout.shape)
    dlow = None if UP_SEPARABLE else torch.empty(tuple(low_shape), dtype=dout.dtype, device=dout.device)
    dskip = torch.empty((N, D, H, W, Cs), dtype=dout.dtype, device=dout.device)
    check(_lib.lib().cbim_upcat_bwd(_dt(dout), _p(dout), _p(dlow), _p(dskip), N, Dl, Hl, Wl, Cl, D, H, W, Cs,
                                    int(skip_first), _stream(dout)), "upcat_bwd")
    if dlow is None:
        dlow = up_adjoint(dout, Cs if skip_first else 0, Cl, (Dl, Hl, Wl))
    return dlow, dskip


# ------------------------------------------------------------------------------------------------
# convolution (implicit GEMM on the matrix cores)
# ------------------------------------------------------------------------------------------------

class ConvGeom:
    """Geometry of one stride-1 nn.Conv3d call: forward and dgrad descriptors."""

    def __init__(self, dtype: torch.dtype, N, in_dhw, Cin, Cout, k, pad, act: int):
        self.k = tuple(int(i) for i in k)
        self.pad = tuple(int(i) for i in pad)
        Di, Hi, Wi = (int(i) for i in in_dhw)
        Do, Ho, Wo = (Di + 2 * self.pad[0] - self.k[0] + 1, Hi + 2 * self.pad[1] - self.k[1] + 1,
                      Wi + 2 * self.pad[2] - self.k[2] + 1)
        dt = 1 if dtype == torch.bfloat16 else 0
        self.N, self.Cin, self.Cout = int(N), int(Cin), int(Cout)
        self.in_dhw, self.out_dhw = (Di, Hi, Wi), (Do, Ho, Wo)
        self.fwd = ConvDesc(dt, N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, *self.k, *self.pad, act)
        dpad = tuple(kk - 1 - pp for kk, pp in zip(self.k, self.pad))
        self.bwd = ConvDesc(dt, N, Do, Ho, Wo, Cout, Di, Hi, Wi, Cin, *self.k, *dpad, act)
        self.dtype = dtype


def pack_weights(w: torch.Tensor, geom: ConvGeom, mode: int) -> torch.Tensor:
    """fp32 [Cout,Cin,kD,kH,kW] -> MFMA B-fragment order in the activation dtype."""
    _dev_ok(w)
    L = _lib.lib()
    nbytes = L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), mode)
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
    check(L.cbim_conv3d_pack_weights(C.byref(geom.fwd), mode, _p(w), _p(packed), _stream(w)), "pack_weights")
    return packed


def pack_weights_both(w: torch.Tensor, geom: ConvGeom):
    """(forward layout, dgrad layout) in one launch."""
    _dev_ok(w)
    L = _lib.lib()
    p0 = torch.empty((L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), 0),), dtype=torch.uint8, device=w.device)
    p1 = torch.empty((L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), 1),), dtype=torch.uint8, device=w.device)
    check(L.cbim_conv3d_pack_weights_both(C.byref(geom.fwd), _p(w), _p(p0), _p(p1), _stream(w)), "pack_weights_both")
    return p0, p1


class _PackEntry:
    __slots__ = ("w0", "w1", "rows0", "geom", "p0", "p1", "versions", "used", "idle", "pinned")


class PackedWeights:
    """Packed (MFMA fragment order) copies of every convolution weight seen so far, re-laid by ONE kernel launch
    when any of them changed (`Tensor._version` moves with the optimizer's in-place update): the first convolution
    of a step finds its weight stale and re-packs the whole table, the others hit.  Replaces one pack launch per
    convolution and the torch.cat of each conv1|shortcut pair (0.5 ms of the 19 ms ResUNet step).  Under hipGraph
    capture the table launch is captured where it happens and replays with the step."""

    MAX_IDLE = 8   # weight-change epochs an entry may go unused before it is dropped

    def __init__(self):
        self.entries = {}
        self.stale = False     # set after a hipGraph replay (the captured optimizer moved the weights unseen)
        self.epoch = getattr(self, "epoch", 0) + 1     # moves with every whole-table re-pack (ops.lo_weights follows it)
        self.table = None      # device copy of the cbim_pack_item array
        self.n_blocks = 0
        self.max_taps = 1
        self.dirty_table = True
        # a hipGraph that captured a table launch replays with the raw pointers of that table and of every entry's
        # p0/p1 baked in: both are kept alive here (and the entries exempt from aging) for the life of the process
        self._graph_refs = []

    @staticmethod
    def _key(ws, geom: ConvGeom):
        return tuple((w.data_ptr(), tuple(w.shape)) for w in ws) + (geom.dtype,)

    def get(self, ws, geom: ConvGeom, need_dgrad: bool):
        """ws: (w,) or (w_conv1, w_shortcut) fp32 [Cout_i, Cin, kD, kH, kW] -> (packed fwd, packed dgrad | None)."""
        key = self._key(ws, geom)
        e = self.entries.get(key)
        if e is None or (need_dgrad and e.p1 is None):
            e = self._add(key, ws, geom, need_dgrad or (e is not None and e.p1 is not None))
        e.used = True
        if self.stale or e.versions != tuple(w._version for w in ws):
            try:
                if e.versions is None and not self.stale:
                    self._pack_one(e)
                else:
                    self._repack_all()
            except Exception:
                if e.versions is None:        # an unsupported weight (e.g. Cout not a multiple of the channel chunk)
                    self.entries.pop(key, None)   # must not stay in the table and fail every later launch
                    self.dirty_table = True
                raise
        return e.p0, (e.p1 if need_dgrad else None)

    def _add(self, key, ws, geom, with_dgrad):
        L = _lib.lib()
        dev = ws[0].device
        e = _PackEntry()
        e.w0 = ws[0].detach()
        e.w1 = ws[1].detach() if len(ws) > 1 else None
        for w in ws:
            _dev_ok(w)
            if w.dtype != torch.float32:
                raise TypeError("cbim_amd: convolution weights must be float32 master copies")
        e.rows0 = int(ws[0].shape[0])
        e.geom = geom
        e.p0 = torch.empty((L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), 0),), dtype=torch.uint8, device=dev)
        e.p1 = torch.empty((L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), 1),), dtype=torch.uint8, device=dev) \
            if with_dgrad else None
        e.versions = None
        e.used, e.idle, e.pinned = True, 0, False
        self.entries[key] = e
        self.dirty_table = True
        return e

    def _build_table(self):
        L = _lib.lib()
        items = (_lib.PackItem * len(self.entries))()
        blk = 0
        for i, e in enumerate(self.entries.values()):
            check(L.cbim_conv3d_pack_item_fill(C.byref(e.geom.fwd), _p(e.w0), _p(e.w1), e.rows0, _p(e.p0), _p(e.p1), blk,
                                               C.byref(items[i])), "pack_item_fill")
            blk += items[i].n_blocks
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        dev = next(iter(self.entries.values())).p0.device
        self.table = raw.to(dev)       # (outside graph capture: see _repack_all)
        self.n_blocks = blk
        self.max_taps = max(int(it.taps) for it in items)
        self.dirty_table = False

    def _pack_one(self, e):
        """a weight seen for the first time: its own launch (the table launch would re-pack every weight)"""
        if e.w1 is None:
            w = e.w0.contiguous()
        else:
            w = torch.cat([e.w0, e.w1], 0).contiguous()
        L = _lib.lib()
        if e.p1 is not None:
            check(L.cbim_conv3d_pack_weights_both(C.byref(e.geom.fwd), _p(w), _p(e.p0), _p(e.p1), _stream(w)), "pack_weights_both")
        else:
            check(L.cbim_conv3d_pack_weights(C.byref(e.geom.fwd), 0, _p(w), _p(e.p0), _stream(w)), "pack_weights")
        e.versions = tuple(t._version for t in ((e.w0,) if e.w1 is None else (e.w0, e.w1)))

    def _repack_all(self):
        dev0 = next(iter(self.entries.values())).p0
        capturing = dev0.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            # a weight CHANGED (optimizer step): entries no convolution asked for during the last 8 such epochs
            # belong to models that are gone
            dead = []
            for k, e in self.entries.items():
                e.idle = 0 if e.used else e.idle + 1
                e.used = False
                if e.idle > self.MAX_IDLE and not e.pinned:
                    dead.append(k)
            for k in dead:
                del self.entries[k]
            self.dirty_table = self.dirty_table or bool(dead)
        if capturing:
            _hook_graph_replay()
        self.stale = False
        self.epoch += 1
        if self.dirty_table:
            if self.table is not None and self.table.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("cbim_amd: a new convolution weight appeared during hipGraph capture; run one "
                                   "eager step first so that the pack table is complete")
            self._build_table()
        t = self.table
        if capturing:
            for e in self.entries.values():
                e.pinned = True
            self._graph_refs.append((t, list(self.entries.values())))
        check(_lib.lib().cbim_conv3d_pack_weights_table(_p(t), len(self.entries), self.n_blocks, self.max_taps, _stream(t)),
              "pack_weights_table")
        for e in self.entries.values():
            e.versions = tuple(w._version for w in ((e.w0,) if e.w1 is None else (e.w0, e.w1)))

    def clear(self):
        self.__init__()

    def release_graphs(self):
        """Call after DESTROYING the hipGraphs that captured table launches (e.g. before re-capturing for another model or input
        shape): drops the tables and packed buffers kept alive for their replays and lets the pinned entries age out like any
        other.  Replaying a graph captured before this call afterwards is undefined (its raw pointers may be gone)."""
        self._graph_refs.clear()
        for e in self.entries.values():
            e.pinned = False


PACKED = PackedWeights()
_REPLAY_HOOKED = False


def _hook_graph_replay():
    """Once a table launch has been captured into a hipGraph, every replay moves the weights (captured optimizer) and
    re-packs them (captured table launch) without this module seeing a `_version` change: an EAGER convolution after a
    replay (validation between graph-replayed epochs) must not trust the host-side versions.  torch offers no replay
    callback, so CUDAGraph.replay is wrapped to flag the cache stale; the next eager lookup then re-packs once."""
    global _REPLAY_HOOKED
    if _REPLAY_HOOKED:
        return
    _REPLAY_HOOKED = True
    orig = torch.cuda.CUDAGraph.replay

    def replay(self):
        orig(self)
        PACKED.stale = True

    torch.cuda.CUDAGraph.replay = replay


def packed_weights(ws, geom: ConvGeom, need_dgrad: bool):
    """(packed forward layout, packed dgrad layout | None) of a convolution weight (or conv1 | shortcut pair).
    nn.Parameters are served from the one-launch-per-optimizer-step table; any other tensor (e.g. the permuted
    ConvTranspose3d weight of monai's UnetrUpBlock, a new allocation every forward pass, with or without grad mode) is
    packed on the spot and never enters the table — entries keep their weight alive, temporaries would pile up."""
    ws = tuple(ws)
    if all(isinstance(w, torch.nn.Parameter) for w in ws):
        return PACKED.get(ws, geom, need_dgrad)
    w = ws[0].detach().contiguous() if len(ws) == 1 else torch.cat([t.detach() for t in ws], 0).contiguous()
    if need_dgrad:
        return pack_weights_both(w, geom)
    return pack_weights(w, geom, 0), None


def conv_igemm(desc: ConvDesc, x, w_packed, out_shape, in_stats=None, res=None, mask_x=None, mask_stats=None,
               want_partials: bool = False, x2=None, out=None):
    """x2: second input tensor; the conv's input is the channel concatenation [x | x2] (never
    materialised), split at x.shape[-1].  out: write into this (dense) tensor instead of a new one; it may be `res`
    itself (in-place accumulation: every output chunk is read and written once, by the same thread)."""
    _dev_ok(x, w_packed, in_stats, res, mask_x, mask_stats, x2, out)
    L = _lib.lib()
    if out is not None:
        assert tuple(out.shape) == tuple(out_shape) and out.dtype == x.dtype and out.is_contiguous()
        y = out
    else:
        y = torch.empty(tuple(out_shape), dtype=x.dtype, device=x.device)
    part = None
    if want_partials:
        tiles = L.cbim_conv3d_num_tiles(C.byref(desc))
        part = torch.empty((desc.N, tiles, desc.Cout, 3), dtype=torch.float32, device=x.device)
    wsb = L.cbim_conv3d_igemm_workspace(C.byref(desc))
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device) if wsb else None
    prof = PROFILE is not None and x.device.type == "cuda"
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.cbim_conv3d_igemm(C.byref(desc), _p(x), _rs(x), _p(x2), _rs(x2) if x2 is not None else 0,
                              int(x.shape[-1]) if x2 is not None else 0, _p(in_stats), _p(w_packed),
                              _p(res), _rs(res) if res is not None else 0,
                              _p(mask_x), _rs(mask_x) if mask_x is not None else 0, _p(mask_stats),
                              _p(y), _rs(y), _p(part), _p(ws), wsb, _stream(x)), "conv3d_igemm")
    if prof:
        e1.record()
        cfg = (C.c_int * 4)()
        L.cbim_conv3d_tile_config(C.byref(desc), C.byref(cfg))
        lk = L.cbim_conv3d_last_kernel()
        if lk == 3:              # low-resolution layers: k_conv3_rw over Cin slices + k_splitk_finish (the row k_conv_igemm<1,2> had)
            name = "k_conv3_rw_splitk<bf16>+finish"
        elif lk == 2:            # conv_rw.hip (raw input: every 3x3x3 launch of the ResUNet step at >= 32^3) — the name rocprofv3 prints
            name = "k_conv3_rw<bf16>"
        elif lk == 1:            # conv_r32.hip (normalise-on-load calls)
            name = "k_conv3_r32<bf16>"
        elif lk == 4:            # 1x1x1 layers: conv_pw.hip
            name = "k_conv_pw<bf16>"
        elif lk == 5:            # round 6: 48 output channels per workgroup (SwinUNETR's feature-48 layers)
            name = "k_conv3_rw48<bf16>"
        else:
            if cfg[0] == 4:       # a shape the r32 kernel takes for other calls: k_conv_igemm runs its 8x8x8 configuration
                cfg[0], cfg[1] = 2, (1 if desc.Cout <= 32 else 2)
            name = "k_conv_igemm<%s,%d,%d>" % ("bf16" if desc.dtype == 1 else "f32", cfg[0], cfg[1])
        flops = 2.0 * desc.N * desc.Do * desc.Ho * desc.Wo * desc.Cout * desc.Cin * desc.kD * desc.kH * desc.kW
        # algorithmic bytes of the launch: every operand tensor read once, the output written once, the packed weights once
        es = x.element_size()
        nbytes = (desc.N * desc.Di * desc.Hi * desc.Wi * desc.Cin * es +
                  desc.N * desc.Do * desc.Ho * desc.Wo * desc.Cout * es * (1 + (res is not None) + (mask_x is not None)) +
                  desc.Cin * desc.Cout * desc.kD * desc.kH * desc.kW * es)
        PROFILE.append((name, flops, e0, e1, (desc.Cin, desc.Cout, desc.Do, desc.Ho, desc.Wo), nbytes))
    return y, part


def conv_fwd(x, w_packed, geom: ConvGeom, in_stats=None, res=None, want_stats=False, eps=IN_EPS):
    """y = conv(act(IN(x))) [+ res]; optionally the InstanceNorm statistics of y (epilogue-fused)."""
    out_shape = (geom.N,) + geom.out_dhw + (geom.Cout,)
    y, part = conv_igemm(geom.fwd, x, w_packed, out_shape, in_stats=in_stats, res=res, want_partials=want_stats)
    stats = None
    if want_stats:
        S = geom.out_dhw[0] * geom.out_dhw[1] * geom.out_dhw[2]
        stats = stats_finalize(part, S, eps, 0)
    return y, stats


def rw48_takes(geom: ConvGeom, masked_dgrad: bool = True) -> bool:
    """True when the engine runs this layer's forward (and, masked_dgrad, its input gradient with an activated mask tensor) on
    k_conv3_rw48 (round 6: Cout in multiples of 48, input used as it is): the caller then materialises act(IN(x)) once."""
    L = _lib.lib()
    if not L.cbim_conv_rw48_takes(C.byref(geom.fwd)):
        return False
    return (not masked_dgrad) or bool(L.cbim_conv_rw48_takes(C.byref(geom.bwd)))


def conv_dgrad(dy, w_packed_dgrad, geom: ConvGeom, mask_x=None, mask_stats=None, accumulate=None, dy2=None):
    """g = dgrad(dy) [+ accumulate] [* act'(xh(mask_x))]; with a mask also returns the two
    InstanceNorm-backward means (m1, m2) computed in the epilogue."""
    out_shape = (geom.N,) + geom.in_dhw + (geom.Cin,)
    g, part = conv_igemm(geom.bwd, dy, w_packed_dgrad, out_shape, res=accumulate, mask_x=mask_x,
                         mask_stats=mask_stats, want_partials=mask_x is not None, x2=dy2)
    sums = None
    if part is not None:
        S = geom.in_dhw[0] * geom.in_dhw[1] * geom.in_dhw[2]
        sums = stats_finalize(part, S, 0.0, 1)
    return g, sums


def slot_of(param):
    """forward side: the gradient-slot handle a data-parallel wrapper attached to `param` (parallel.GradAllReduce), or None"""
    return getattr(param, "_cbim_grad_slot", None) if param is not None else None


def grad_slot(handle):
    """backward side: the slot of the wrapper's flat gradient bucket behind `handle`, claimed for the running backward pass: a
    weight-gradient kernel given this tensor as `out` writes the gradient where the collective reads it — no copy into the
    bucket.  None: no wrapper, the slot was already written in this pass (shared weight), or the parameter holds an
    accumulated gradient."""
    return None if handle is None else handle.claim()


def grad_slot_pair(h1, h2):
    """the slots behind two handles as one Cout-concatenated tensor (parallel._GradSlot.claim_with) -> (cat, alias1, alias2) or None"""
    return None if h1 is None or h2 is None else h1.claim_with(h2)


def _dw_out(out, shape, device):
    if out is None:
        return torch.empty(tuple(shape), dtype=torch.float32, device=device)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(shape), "cbim_amd: bad gradient slot"
    return out


WGRAD_EMBED_133 = True   # (1,3,3) weight gradients as the centre plane of a 3x3x3 one on k_wgrad_r32 (tests switch it off for A/Bs)


def conv_wgrad(x, in_stats, dy, geom: ConvGeom, dy2=None, x2=None, out=None) -> torch.Tensor:
    """dy2: gradient of the output channels >= dy.shape[-1] (Cout-concatenated convs); x2: the input channels
    >= x.shape[-1] (virtual concatenation, raw bf16 3x3x3 inputs only); out: where to write (ops.grad_slot)."""
    _dev_ok(x, in_stats, dy, dy2, x2)
    if (WGRAD_EMBED_133 and geom.k == (1, 3, 3) and geom.pad == (0, 1, 1) and in_stats is None and x.dtype == torch.bfloat16
            and geom.Cin % 32 == 0 and geom.Cout % 32 == 0 and min(geom.in_dhw) >= 8):
        # round 6: the (1, 3, 3) kernels of the ACDC-structured configurations (config/acdc/*.yaml: kernel_size [[1,3,3],[1,3,3],...]) —
        # their weight gradient is the CENTRE PLANE of the 3x3x3 weight gradient of the same tensors, which k_wgrad_r32 computes in
        # about half the time k_conv_wgrad needs for the nine taps (72 -> 34 us on 32 -> 32 over 16x192x192) despite 3 x the MFMAs
        g3 = ConvGeom(x.dtype, geom.N, geom.in_dhw, geom.Cin, geom.Cout, (3, 3, 3), (1, 1, 1), 0)
        centre = conv_wgrad(x, None, dy, g3, dy2=dy2, x2=x2)[:, :, 1:2]
        if out is None:
            return centre.contiguous()
        out.copy_(centre)
        return out
    L = _lib.lib()
    nbytes = L.cbim_conv3d_wgrad_workspace(C.byref(geom.fwd))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    dw = _dw_out(out, (geom.Cout, geom.Cin) + geom.k, x.device)
    prof = PROFILE is not None and x.device.type == "cuda"
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.cbim_conv3d_wgrad(C.byref(geom.fwd), _p(x), _rs(x), _p(x2), _rs(x2) if x2 is not None else 0,
                              int(x.shape[-1]) if x2 is not None else 0, _p(in_stats), _p(dy), _rs(dy),
                              _p(dy2), _rs(dy2) if dy2 is not None else 0,
                              int(dy.shape[-1]) if dy2 is not None else 0,
                              _p(dw), _p(ws), nbytes, _stream(x)), "conv3d_wgrad")
    if prof:
        e1.record()
        d = geom.fwd
        flops = 2.0 * d.N * d.Do * d.Ho * d.Wo * d.Cout * d.Cin * d.kD * d.kH * d.kW
        lkw = L.cbim_conv3d_wgrad_last_kernel()
        name = "k_wgrad_r32<bf16>+reduce" if lkw == 1 else "k_pw_wgrad<bf16>+reduce" if lkw == 2 else \
            "k_conv_wgrad<%s>+reduce" % ("bf16" if d.dtype == 1 else "f32")
        nbytes = d.N * d.Do * d.Ho * d.Wo * (d.Cin + d.Cout) * x.element_size() + dw.numel() * 4
        PROFILE.append((name, flops, e0, e1,
                        (d.Cin, d.Cout, d.Do, d.Ho, d.Wo), nbytes))
    return dw


# ------------------------------------------------------------------------------------------------
# stem / head
# ------------------------------------------------------------------------------------------------

def stem_fwd(x_ncdhw: torch.Tensor, w: torch.Tensor, pad, out_dtype: torch.dtype):
    _dev_ok(x_ncdhw, w)
    N, Cin, Di, Hi, Wi = map(int, x_ncdhw.shape)
    Cout, _, kD, kH, kW = map(int, w.shape)
    pD, pH, pW = map(int, pad)
    Do, Ho, Wo = Di + 2 * pD - kD + 1, Hi + 2 * pH - kH + 1, Wi + 2 * pW - kW + 1
    y = torch.empty((N, Do, Ho, Wo, Cout), dtype=out_dtype, device=x_ncdhw.device)
    check(_lib.lib().cbim_stem_conv_fwd(_dt(y), _p(x_ncdhw), _p(w), _p(y), N, Cin, Di, Hi, Wi, Cout, kD, kH, kW,
                                        pD, pH, pW, Do, Ho, Wo, _stream(y)), "stem_conv_fwd")
    return y


def stem_wgrad(x_ncdhw, dy, w_shape, pad, out=None):
    _dev_ok(x_ncdhw, dy)
    N, Cin, Di, Hi, Wi = map(int, x_ncdhw.shape)
    Cout, _, kD, kH, kW = map(int, w_shape)
    pD, pH, pW = map(int, pad)
    _, Do, Ho, Wo, _ = map(int, dy.shape)
    L = _lib.lib()
    nbytes = L.cbim_stem_conv_wgrad_workspace(N, Cin, Cout, kD, kH, kW, Do, Ho, Wo)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dy.device)
    dw = _dw_out(out, w_shape, dy.device)
    check(L.cbim_stem_conv_wgrad(_dt(dy), _p(x_ncdhw), _p(dy), _p(dw), N, Cin, Di, Hi, Wi, Cout, kD, kH, kW,
                                 pD, pH, pW, Do, Ho, Wo, _p(ws), nbytes, _stream(dy)), "stem_conv_wgrad")
    return dw


def head_fwd(x, w2d, b):
    """x [N,D,H,W,Cin] -> logits float32 [N,K,D,H,W]; w2d float32 [K,Cin]."""
    _dev_ok(x, w2d, b)
    N, D, H, W, Cin = map(int, x.shape)
    K = int(w2d.shape[0])
    logits = torch.empty((N, K, D, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().cbim_head_fwd(_dt(x), _p(x), _p(w2d), _p(b), _p(logits), N, D * H * W, Cin, K, _stream(x)),
          "head_fwd")
    return logits


def head_bwd(x, w2d, dlogits, need_dx=True, out_w=None, out_b=None):
    _dev_ok(x, w2d, dlogits)
    N, D, H, W, Cin = map(int, x.shape)
    K = int(w2d.shape[0])
    S = D * H * W
    L = _lib.lib()
    nbytes = L.cbim_head_bwd_workspace(S, N, Cin, K)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    dw = _dw_out(out_w, (K, Cin), x.device)
    db = _dw_out(out_b, (K,), x.device)
    check(L.cbim_head_bwd(_dt(x), _p(x), _p(w2d), _p(dlogits), _p(dx), _p(dw), _p(db), N, S, Cin, K, _p(ws), nbytes,
                          _stream(x)), "head_bwd")
    return dx, dw, db


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------

def dice_ce_fwd(logits, labels, weight=None):
    """-> out float32[4] = (CE, Dice, CE+Dice, #labels outside [0,C)), coef float32[3C+1] (2C+1 backward coefficients, then the
    C per-class terms 1 - dice_c)."""
    _dev_ok(logits, labels, weight)
    N, Cc = int(logits.shape[0]), int(logits.shape[1])
    S = logits.numel() // (N * Cc)
    L = _lib.lib()
    nbytes = L.cbim_dice_ce_workspace(N, Cc, S)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=logits.device)
    out = torch.empty((4,), dtype=torch.float32, device=logits.device)
    coef = torch.empty((3 * Cc + 1,), dtype=torch.float32, device=logits.device)    # [2C + 1] backward coefficients + [C] per-class terms
    check(L.cbim_dice_ce_fwd(_p(logits), _p(labels), _p(weight), N, Cc, S, _p(out), _p(coef), _p(ws), nbytes,
                             _stream(logits)), "dice_ce_fwd")
    return out, coef


def dice_ce_bwd(logits, labels, weight, coef, grad2):
    _dev_ok(logits, labels, weight, coef, grad2)
    N, Cc = int(logits.shape[0]), int(logits.shape[1])
    S = logits.numel() // (N * Cc)
    dz = torch.empty_like(logits)
    check(_lib.lib().cbim_dice_ce_bwd(_p(logits), _p(labels), _p(weight), _p(coef), _p(grad2), _p(dz), N, Cc, S,
                                      _stream(logits)), "dice_ce_bwd")
    return dz


def ncdhw_to_ndhwc(x: torch.Tensor, dtype: torch.dtype):
    _dev_ok(x)
    N, Cc = int(x.shape[0]), int(x.shape[1])
    S = x.numel() // (N * Cc)
    y = torch.empty((N,) + tuple(x.shape[2:]) + (Cc,), dtype=dtype, device=x.device)
    check(_lib.lib().cbim_ncdhw_to_ndhwc(_dt(y), _p(x), _p(y), N, Cc, S, _stream(x)), "ncdhw_to_ndhwc")
    return y


def ndhwc_to_ncdhw(x: torch.Tensor):
    _dev_ok(x)
    N, Cc = int(x.shape[0]), int(x.shape[-1])
    S = x.numel() // (N * Cc)
    y = torch.empty((N, Cc) + tuple(x.shape[1:-1]), dtype=torch.float32, device=x.device)
    check(_lib.lib().cbim_ndhwc_to_ncdhw(_dt(x), _p(x), _p(y), N, Cc, S, _stream(x)), "ndhwc_to_ncdhw")
    return y


# ------------------------------------------------------------------------------------------------
# MedFormer pieces
# ------------------------------------------------------------------------------------------------

def dwconv(x, w2d, k, in_stats=None, act: int = 0, bias=None, flip: bool = False):
    """depthwise conv, stride 1, pad k//2; w2d float32 [C, kD*kH*kW]; optional fused IN(+act) on load."""
    _dev_ok(x, w2d, in_stats, bias)
    N, D, H, W, Cc = map(int, x.shape)
    y = torch.empty((N, D, H, W, Cc), dtype=x.dtype, device=x.device)
    check(_lib.lib().cbim_dwconv3d(_dt(x), _p(x), _rs(x), _p(in_stats), act, _p(bias), _p(w2d), int(flip), _p(y), Cc,
                                   N, D, H, W, Cc, int(k[0]), int(k[1]), int(k[2]), _stream(x)), "dwconv3d")
    return y


DW_WGRAD_MFMA = True   # depthwise wgrad on k_wgrad_r32 (diagonal of 32-channel groups)


def dwconv_wgrad_on_matrix_cores(x, k) -> bool:
    """Whether cbim_dwconv3d_wgrad takes the matrix-core form for this tensor when it gets the input AS IT IS and no
    gradient bias (the caller then materialises act(IN(x)) and dy + bias first)."""
    return (DW_WGRAD_MFMA and x.dtype == torch.bfloat16 and tuple(int(i) for i in k) == (3, 3, 3) and int(x.shape[-1]) % 32 == 0
            and min(int(x.shape[1]), int(x.shape[2]), int(x.shape[3])) >= 8)


_SHIFT_CONSTS = {}


def _shift_consts(device):
    """([-1, 0], [0, 1]) on `device`, built once (outside any graph capture: the first eager step)"""
    key = str(device)
    if key not in _SHIFT_CONSTS:
        _SHIFT_CONSTS[key] = (torch.tensor([-1.0, 0.0], device=device), torch.tensor([0.0, 1.0], device=device))
    return _SHIFT_CONSTS[key]


def dwconv_wgrad(x, in_stats, act: int, dy, k, dy_bias=None):
    _dev_ok(x, in_stats, dy, dy_bias)
    N, D, H, W, Cc = map(int, x.shape)
    kD, kH, kW = (int(i) for i in k)
    L = _lib.lib()
    if dwconv_wgrad_on_matrix_cores(x, k) and (in_stats is not None or dy_bias is not None):
        # one streaming pass each: a = act(IN(x)) and dy + bias (fp32 add, rounded once — what autograd's bf16 accumulation of
        # the two gradient branches does in the reference under autocast), then the raw form
        if in_stats is not None:
            x = norm_act_fwd(x, in_stats, act)
        if dy_bias is not None:
            # (mean, rstd) = (-bias, 1) per (n, c): dy + bias through the normalisation kernel; ONE broadcast launch builds the
            # pairs (was neg / ones_like / stack: three launches per MBConv backward)
            c, d = _shift_consts(dy_bias.device)
            shift = torch.addcmul(d, dy_bias.float().unsqueeze(-1), c)
            dy = norm_act_fwd(dy, shift, ACT["none"])
        in_stats, dy_bias, act = None, None, 0
    nbytes = L.cbim_dwconv3d_wgrad_workspace(N, D, H, W, Cc, kD, kH, kW)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    dw = torch.empty((Cc, kD * kH * kW), dtype=torch.float32, device=x.device)
    check(L.cbim_dwconv3d_wgrad(_dt(x), _p(x), _rs(x), _p(in_stats), act, _p(dy), _rs(dy), _p(dy_bias), _p(dw),
                                N, D, H, W, Cc, kD, kH, kW, _p(ws), nbytes, _stream(x)), "dwconv3d_wgrad")
    return dw


def space_to_depth(x, scale):
    _dev_ok(x)
    N, D, H, W, Cc = map(int, x.shape)
    sD, sH, sW = (int(i) for i in scale)
    y = torch.empty((N, D // sD, H // sH, W // sW, Cc * sD * sH * sW), dtype=x.dtype, device=x.device)
    check(_lib.lib().cbim_space_to_depth(_dt(x), _p(x), _p(y), N, D, H, W, Cc, sD, sH, sW, 0, _stream(x)),
          "space_to_depth")
    return y


def depth_to_space(dy, in_shape, scale):
    _dev_ok(dy)
    N, D, H, W, Cc = map(int, in_shape)
    sD, sH, sW = (int(i) for i in scale)
    dx = torch.empty(tuple(in_shape), dtype=dy.dtype, device=dy.device)
    check(_lib.lib().cbim_space_to_depth(_dt(dy), _p(dy), _p(dx), N, D, H, W, Cc, sD, sH, sW, 1, _stream(dy)),
          "space_to_depth(inverse)")
    return dx


def depth_to_space_into(t, out, Cc: int, scale):
    """scatter t [N, D/s, H/s, W/s, s^3 Cc] into the first Cc channels of the wider channels-last tensor out [N, D, H, W, >= Cc]"""
    _dev_ok(t, out)
    N, D, H, W = (int(v) for v in out.shape[:4])
    sD, sH, sW = (int(i) for i in scale)
    check(_lib.lib().cbim_space_to_depth_strided(_dt(t), _p(t), _p(out), N, D, H, W, Cc, sD, sH, sW, 1, int(out.stride(3)), _stream(t)),
          "space_to_depth_strided(inverse)")
    return out


def space_to_depth_from(g, Cc: int, scale):
    """gather the first Cc channels of the channels-last (row-strided) tensor g [N, D, H, W, >= Cc] into [N, D/s, H/s, W/s, s^3 Cc]"""
    _dev_ok(g)
    N, D, H, W = (int(v) for v in g.shape[:4])
    sD, sH, sW = (int(i) for i in scale)
    y = torch.empty((N, D // sD, H // sH, W // sW, Cc * sD * sH * sW), dtype=g.dtype, device=g.device)
    check(_lib.lib().cbim_space_to_depth_strided(_dt(g), _p(g), _p(y), N, D, H, W, Cc, sD, sH, sW, 0, int(g.stride(3)), _stream(g)),
          "space_to_depth_strided")
    return y


def _attn_ws(N, L, heads, dh, M, dev):
    nbytes = _lib.lib().cbim_bidir_attn_workspace(N, L, heads, dh, M)
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


def bidir_attn_fwd(qv, mq, mv, heads: int, scale: float):
    """qv [N,D,H,W,2*inner]; mq, mv float32 [N,M,inner] -> feat_out [N,D,H,W,inner], map_out [N,M,inner], colstat."""
    _dev_ok(qv, mq, mv)
    N, Lr, inner = int(qv.shape[0]), _spatial(qv), int(qv.shape[-1]) // 2
    M, dh = int(mq.shape[1]), inner // heads
    fo = torch.empty(tuple(qv.shape[:-1]) + (inner,), dtype=qv.dtype, device=qv.device)
    mo = torch.empty((N, M, inner), dtype=torch.float32, device=qv.device)
    cs = torch.empty((N, heads, M, 2), dtype=torch.float32, device=qv.device)
    ws, nbytes = _attn_ws(N, Lr, heads, dh, M, qv.device)
    check(_lib.lib().cbim_bidir_attn_fwd(_dt(qv), _p(qv), _rs(qv), _p(mq), _p(mv), _p(fo), _p(mo), _p(cs), N, Lr, heads,
                                         dh, M, float(scale), _p(ws), nbytes, _stream(qv)), "bidir_attn_fwd")
    return fo, mo, cs


def bidir_attn_bwd(qv, mq, mv, cs, mo, dfo, dmo, heads: int, scale: float):
    _dev_ok(qv, mq, mv, cs, mo, dfo, dmo)
    N, Lr, inner = int(qv.shape[0]), _spatial(qv), int(qv.shape[-1]) // 2
    M, dh = int(mq.shape[1]), inner // heads
    dqv = torch.empty(tuple(qv.shape), dtype=qv.dtype, device=qv.device)
    dmq = torch.empty((N, M, inner), dtype=torch.float32, device=qv.device)
    dmv = torch.empty((N, M, inner), dtype=torch.float32, device=qv.device)
    ws, nbytes = _attn_ws(N, Lr, heads, dh, M, qv.device)
    check(_lib.lib().cbim_bidir_attn_bwd(_dt(qv), _p(qv), _rs(qv), _p(mq), _p(mv), _p(cs), _p(mo), _p(dfo), _p(dmo),
                                         _p(dqv), _p(dmq), _p(dmv), N, Lr, heads, dh, M, float(scale), _p(ws), nbytes,
                                         _stream(qv)), "bidir_attn_bwd")
    return dqv, dmq, dmv


def map_gemm(A, X, OUT, O: int, K: int, Nn: int, batch: int, *, lda, ldx, ldo, a_t=0, x_t=0, o_t=0, A2=None, a_split=0, X2=None,
             x_split=0, OUT2=None, o_split=0, R=None, ldr=0, a_batch=0, x_batch=0, o_batch=0, r_batch=0, reduce_batch=0,
             ln_eps=None, Xn=None, rstd_out=None, XH=None, rstd_in=None):
    """OUT[o][n] = sum_k A[o][k] X[k][n] (+ R) in float32 on the engine's small-GEMM kernel (map_kernels.hip; operand layouts and
    the fused InstanceNorm steps: include/cbim_hip.h cbim_map_gemm_desc) — MedFormer's semantic-map branch."""
    ts = [t for t in (A, A2, X, X2, OUT, OUT2, R, Xn, rstd_out, XH, rstd_in) if t is not None]
    _dev_ok(*ts)
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in ts), "cbim_amd: map_gemm takes contiguous float32 tensors"
    d = _lib.MapGemmDesc()
    d.A, d.A2, d.lda, d.a_batch, d.a_t, d.a_split = _p(A), _p(A2), lda, a_batch, a_t, a_split
    d.X, d.X2, d.ldx, d.x_batch, d.x_t, d.x_split = _p(X), _p(X2), ldx, x_batch, x_t, x_split
    d.OUT, d.OUT2, d.ldo, d.o_batch, d.o_t, d.o_split = _p(OUT), _p(OUT2), ldo, o_batch, o_t, o_split
    d.R, d.ldr, d.r_batch = _p(R), ldr, r_batch
    d.O, d.K, d.Nn, d.batch, d.reduce_batch, d.ln_mode = O, K, Nn, batch, reduce_batch, int(ln_eps is not None)
    d.eps = float(ln_eps) if ln_eps is not None else 0.0
    d.Xn, d.rstd_out, d.XH, d.rstd_in = _p(Xn), _p(rstd_out), _p(XH), _p(rstd_in)
    check(_lib.lib().cbim_map_gemm(C.byref(d), _stream(A)), "map_gemm")
    return OUT


MAP_MAX_POSITIONS = 128     # columns of one k_map_gemm tile: the normalisation steps need a map row inside it


def map_qv_fwd(smap, wqv, eps: float):
    """norm2 + map_qv of a BidirectionAttentionBlock: smap [B, C, M], wqv [2I, C] -> (mq [B, M, I], mv [B, M, I], normalised map
    [B, C, M], rstd [B, C])."""
    B, Cm, M = (int(v) for v in smap.shape)
    I = int(wqv.shape[0]) // 2
    mq = torch.empty((B, M, I), dtype=torch.float32, device=smap.device)
    mv = torch.empty_like(mq)
    mapp = torch.empty_like(smap)
    rstd = torch.empty((B, Cm), dtype=torch.float32, device=smap.device)
    map_gemm(wqv, smap, mq, 2 * I, Cm, M, B, lda=Cm, ldx=M, ldo=I, o_t=1, OUT2=mv, o_split=I, x_batch=Cm * M, o_batch=M * I,
             ln_eps=eps, Xn=mapp, rstd_out=rstd)
    return mq, mv, mapp, rstd


def map_qv_bwd(dmq, dmv, wqv, mapp, rstd, need_dw=True):
    """-> (d smap [B, C, M] through map_qv and norm2, d wqv [2I, C])."""
    B, Cm, M = (int(v) for v in mapp.shape)
    I = int(wqv.shape[0]) // 2
    dw = None
    if need_dw:
        dw = torch.empty((2 * I, Cm), dtype=torch.float32, device=mapp.device)
        map_gemm(dmq, mapp, dw, 2 * I, M, Cm, B, lda=I, ldx=M, ldo=Cm, a_t=1, A2=dmv, a_split=I, x_t=1, a_batch=M * I, x_batch=Cm * M,
                 reduce_batch=1)
    ds = torch.empty_like(mapp)
    map_gemm(wqv, dmq, ds, Cm, 2 * I, M, B, lda=Cm, ldx=I, ldo=M, a_t=1, x_t=1, X2=dmv, x_split=I, x_batch=M * I, o_batch=Cm * M,
             XH=mapp, rstd_in=rstd)
    return ds, dw


def map_out_fwd(mo, wout, smap):
    """map_out projection + residual: mo [B, M, I], wout [C, I], smap [B, C, M] -> [B, C, M]."""
    B, M, I = (int(v) for v in mo.shape)
    Cm = int(wout.shape[0])
    out = torch.empty((B, Cm, M), dtype=torch.float32, device=mo.device)
    map_gemm(wout, mo, out, Cm, I, M, B, lda=I, ldx=I, ldo=M, x_t=1, x_batch=M * I, o_batch=Cm * M, R=smap, ldr=M, r_batch=Cm * M)
    return out


def map_out_bwd(g, wout, mo, need_dw=True):
    """g [B, C, M] -> (d mo [B, M, I], d wout [C, I])."""
    B, M, I = (int(v) for v in mo.shape)
    Cm = int(wout.shape[0])
    dmo = torch.empty_like(mo)
    map_gemm(wout, g, dmo, I, Cm, M, B, lda=I, ldx=M, ldo=I, a_t=1, o_t=1, x_batch=Cm * M, o_batch=M * I)
    dw = None
    if need_dw:
        dw = torch.empty((Cm, I), dtype=torch.float32, device=mo.device)
        map_gemm(g, mo, dw, Cm, M, I, B, lda=M, ldx=I, ldo=I, a_batch=Cm * M, x_batch=M * I, reduce_batch=1)
    return dmo, dw


def se_gate_fwd(mean, w1, b1, w2, b2):
    """SEBlock.excitation on the channel means: mean [N, C], w1 [H, C], w2 [C, H] (float32) -> (gate [N, C], z1 [N, H])."""
    ts = [t for t in (mean, w1, b1, w2, b2) if t is not None]
    _dev_ok(*ts)
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in ts), "cbim_amd: se_gate takes contiguous float32 tensors"
    N, Cc = (int(v) for v in mean.shape)
    H = int(w1.shape[0])
    z1 = torch.empty((N, H), dtype=torch.float32, device=mean.device)
    gate = torch.empty((N, Cc), dtype=torch.float32, device=mean.device)
    check(_lib.lib().cbim_se_gate_fwd(_p(mean), _p(w1), _p(b1), _p(w2), _p(b2), _p(z1), _p(gate), N, Cc, H, _stream(mean)), "se_gate_fwd")
    return gate, z1


def se_gate_bwd(dgate, gate, z1, mean, w1, w2, need_dmean=True, need_bias=(True, True)):
    _dev_ok(dgate, gate, z1, mean, w1, w2)
    N, Cc = (int(v) for v in mean.shape)
    H = int(w1.shape[0])
    dev = mean.device
    dz1 = torch.empty((N, H), dtype=torch.float32, device=dev)
    dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
    db1 = torch.empty((H,), dtype=torch.float32, device=dev) if need_bias[0] else None
    db2 = torch.empty((Cc,), dtype=torch.float32, device=dev) if need_bias[1] else None
    dmean = torch.empty_like(mean) if need_dmean else None
    check(_lib.lib().cbim_se_gate_bwd(_p(dgate), _p(gate), _p(z1), _p(mean), _p(w1), _p(w2), _p(dz1), _p(dw1), _p(db1), _p(dw2), _p(db2),
                                      _p(dmean), N, Cc, H, _stream(mean)), "se_gate_bwd")
    return dmean, dw1, db1, dw2, db2


AWG_ROWS = 256        # include/cbim_hip.h CBIM_AWG_ROWS: rows of S per column record
AWG = True            # the attention core as matrix products where the register-resident kernels do not apply (tests switch it off for A/Bs)


def awg_eligible(qv, mq, heads: int) -> bool:
    """Whether the BidirectionAttention core of this call runs as matrix products (functional.BidirAttnFn): bf16 feature rows, a head /
    map size the register-resident kernels do not take (d_head not in 8 | 16 | 32, or more than 64 codes — csrc/medformer_kernels.hip
    attn_wide) and that the softmax kernels of csrc/attn_gemm_kernels.hip do: 1 | 2 | 4 | 8 heads, codes in multiples of 8 up to 128
    (config/lits: one head of 128 / 256 / 320 channels, 64 codes; config/acdc: 4 heads, 72 codes)."""
    inner, M = int(qv.shape[-1]) // 2, int(mq.shape[1])
    if not (AWG and qv.dtype == torch.bfloat16 and heads in (1, 2, 4, 8) and inner % heads == 0 and inner % 8 == 0 and inner >= 64):
        return False
    dh = inner // heads
    wide = dh not in (8, 16, 32) or M > 64
    lpr = 4 if M <= 32 else 8 if M <= 64 else 16
    return wide and M % 8 == 0 and 8 <= M <= 128 and heads <= 64 // lpr and (heads * M) % 8 == 0 and _spatial(qv) >= 512


def awg_rows(S, heads: int, scale: float):
    """S float32 [L, H*M] -> (P bf16 [L, H*M] = softmax over the M codes of every (voxel, head), column records)"""
    _dev_ok(S)
    Lr, M = int(S.shape[0]), int(S.shape[1]) // heads
    P = torch.empty((Lr, heads * M), dtype=torch.bfloat16, device=S.device)
    rec = torch.empty(((Lr * heads + AWG_ROWS - 1) // AWG_ROWS, heads, M, 2), dtype=torch.float32, device=S.device)
    check(_lib.lib().cbim_awg_rows(_p(S), Lr, heads, M, float(scale), _p(P), _p(rec), _stream(S)), "awg_rows")
    return P, rec


def awg_cols(S, heads: int, scale: float, rec):
    """-> (Cs bf16 [L, H*M] = softmax over the L voxels of every (head, code), lse float32 [H, M])"""
    _dev_ok(S, rec)
    Lr, M = int(S.shape[0]), int(S.shape[1]) // heads
    Cs = torch.empty((Lr, heads * M), dtype=torch.bfloat16, device=S.device)
    lse = torch.empty((heads, M), dtype=torch.float32, device=S.device)
    check(_lib.lib().cbim_awg_cols(_p(S), Lr, heads, M, float(scale), _p(rec), _p(lse), _p(Cs), _stream(S)), "awg_cols")
    return Cs, lse


def awg_ds(dP, P, dC, Cs, dmo, mo, heads: int, scale: float):
    """dS bf16 [L, H*M] = scale (P o (dP - rowsum(dP o P)) + C o (dC - colsum)), colsum[h][m] = sum over the head's channels of dmo o mo"""
    _dev_ok(dP, P, dC, Cs, dmo, mo)
    Lr, M, inner = int(dP.shape[0]), int(dP.shape[1]) // heads, int(mo.shape[1])
    dS = torch.empty((Lr, heads * M), dtype=torch.bfloat16, device=dP.device)
    ws = torch.empty((heads * M,), dtype=torch.float32, device=dP.device)
    check(_lib.lib().cbim_awg_ds(_p(dP), _p(P), _p(dC), _p(Cs), _p(dmo), _p(mo), inner, Lr, heads, M, float(scale), _p(ws), _p(dS),
                                 _stream(dP)), "awg_ds")
    return dS


def _pack_lin(w2d):
    """packed forward image of a float32 [Cout, Cin] matrix used as the weight of a row GEMM"""
    w2d = w2d.contiguous()
    g = linear_geom(int(w2d.shape[1]), int(w2d.shape[0]))
    return pack_weights(w2d.view(w2d.shape[0], w2d.shape[1], 1, 1, 1), g, 0)


_HEAD_MASKS = {}


def _head_mask(heads: int, inner: int, device):
    """float32 [H, 1, inner]: 1 where channel c belongs to head h (c % H == h — '(dim_head heads)', medformer_utils.py:43-59); built once"""
    key = (heads, inner, str(device))
    m = _HEAD_MASKS.get(key)
    if m is None:
        c = torch.arange(inner) % heads
        m = _HEAD_MASKS[key] = (c[None, :] == torch.arange(heads)[:, None]).float().view(heads, 1, inner).to(device)
    return m


def _blocks(w, heads: int):
    """map-side rows [M, inner] -> the block matrix of all heads [H*M, inner]: row h*M + m keeps the channels of head h"""
    if heads == 1:
        return w
    return (w.unsqueeze(0) * _head_mask(heads, int(w.shape[1]), w.device)).reshape(heads * int(w.shape[0]), int(w.shape[1]))


def _diag(G, heads: int):
    """the inverse gather: G [H*M, inner] (a product over all head pairs) -> [M, inner], entry (m, c) from row (c % H)*M + m"""
    if heads == 1:
        return G
    M = int(G.shape[0]) // heads
    return (G.view(heads, M, int(G.shape[1])) * _head_mask(heads, int(G.shape[1]), G.device)).sum(0)


def bidir_attn_gemm_fwd(qv, mq, mv, heads: int, scale: float):
    """BidirectionAttention core as matrix products (medformer_utils.py:63-97; csrc/attn_gemm_kernels.hip): per image
    S = Q MQ^T of all heads in ONE row GEMM whose weight is the block matrix of the heads (fp32 out, [L, H*M]) -> P, C (both softmaxes,
    bf16) -> feat_out = P MV (row GEMM), map_out = the head-diagonal of C^T FV (weight-gradient GEMM over the voxels).  The products
    between different heads that the block form computes and discards are a few GFLOP at most.
    Returns (feat_out like qv[..., :inner], map_out float32 [N, M, inner], P, C)."""
    N, Lr, inner = int(qv.shape[0]), _spatial(qv), int(qv.shape[-1]) // 2
    M = int(mq.shape[1])
    HM = heads * M
    rows = qv.reshape(N, Lr, 2 * inner)
    fo = torch.empty(tuple(qv.shape[:-1]) + (inner,), dtype=qv.dtype, device=qv.device)
    fo_r = fo.view(N, Lr, inner)
    mo = torch.empty((N, M, inner), dtype=torch.float32, device=qv.device)
    Ps, Cs = [], []
    for n in range(N):
        q, v = rows[n, :, :inner], rows[n, :, inner:]
        S = token_linear(q, _pack_lin(_blocks(mq[n], heads)), None, HM, out_dtype=torch.float32)   # [L, H*M]
        P, rec = awg_rows(S, heads, scale)
        Cn, _ = awg_cols(S, heads, scale, rec)
        token_linear(P, _pack_lin(_blocks(mv[n], heads).t()), None, inner, out=fo_r[n])            # P [L, H*M] x MV blocks [H*M, inner]
        if heads == 1:
            token_linear_wgrad(v, Cn, out=mo[n])                                                   # sum_l C[l][m] v[l][d]
        else:
            mo[n] = _diag(token_linear_wgrad(v, Cn), heads)
        Ps.append(P)
        Cs.append(Cn)
    return fo, mo, torch.stack(Ps), torch.stack(Cs)


def bidir_attn_gemm_bwd(qv, mq, mv, P, Cs, mo, dfo, dmo, heads: int, scale: float):
    N, Lr, inner = int(qv.shape[0]), _spatial(qv), int(qv.shape[-1]) // 2
    M = int(mq.shape[1])
    HM = heads * M
    rows = qv.reshape(N, Lr, 2 * inner)
    dfo_r = dfo.reshape(N, Lr, inner)
    dqv = torch.empty(tuple(qv.shape), dtype=qv.dtype, device=qv.device)
    dq_r = dqv.view(N, Lr, 2 * inner)
    dmq = torch.empty((N, M, inner), dtype=torch.float32, device=qv.device)
    dmv = torch.empty((N, M, inner), dtype=torch.float32, device=qv.device)
    for n in range(N):
        q, v = rows[n, :, :inner], rows[n, :, inner:]
        bq, bv, bd = _blocks(mq[n], heads), _blocks(mv[n], heads), _blocks(dmo[n], heads)
        dP = token_linear(dfo_r[n], _pack_lin(bv), None, HM, out_dtype=torch.float32)             # dfo [L, inner] x MV^T
        dC = token_linear(v, _pack_lin(bd), None, HM, out_dtype=torch.float32)                     # FV [L, inner] x dmo^T
        token_linear(Cs[n], _pack_lin(bd.t()), None, inner, out=dq_r[n, :, inner:])                # dFV = C dmo
        dS = awg_ds(dP, P[n], dC, Cs[n], dmo[n], mo[n], heads, scale)
        token_linear(dS, _pack_lin(bq.t()), None, inner, out=dq_r[n, :, :inner])                   # dQ = dS MQ
        if heads == 1:
            token_linear_wgrad(dfo_r[n], P[n], out=dmv[n])                                         # sum_l P[l][m] dfo[l][d]
            token_linear_wgrad(q, dS, out=dmq[n])                                                  # dMQ = dS^T Q
        else:
            dmv[n] = _diag(token_linear_wgrad(dfo_r[n], P[n]), heads)
            dmq[n] = _diag(token_linear_wgrad(q, dS), heads)
    return dqv, dmq, dmv


def colsoftmax_pool_fwd(fw, Cf: int):
    """fw [N,D,H,W,Cf+M] -> map float32 [N,Cf,M], colstat float32 [N,M,2]."""
    _dev_ok(fw)
    N, Lr, M = int(fw.shape[0]), _spatial(fw), int(fw.shape[-1]) - Cf
    L = _lib.lib()
    nbytes = L.cbim_colsoftmax_pool_workspace(N, Lr, Cf, M)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=fw.device)
    mp = torch.empty((N, Cf, M), dtype=torch.float32, device=fw.device)
    cs = torch.empty((N, M, 2), dtype=torch.float32, device=fw.device)
    check(L.cbim_colsoftmax_pool_fwd(_dt(fw), _p(fw), _rs(fw), _p(mp), _p(cs), N, Lr, Cf, M, _p(ws), nbytes, _stream(fw)),
          "colsoftmax_pool_fwd")
    return mp, cs


MAPPOOL_BWD_GEMM = True   # backward as two library GEMMs (off: k_mappool_bwd4)
MAPPOOL_BWD_GEMM_MAX_L = 1024   # voxels per image up to which it is used


def colsoftmax_pool_bwd(fw, Cf: int, mp, cs, dmap):
    _dev_ok(fw, mp, cs, dmap)
    N, Lr, M = int(fw.shape[0]), _spatial(fw), int(fw.shape[-1]) - Cf
    if MAPPOOL_BWD_GEMM and Lr <= MAPPOOL_BWD_GEMM_MAX_L:
        # the backward of map[c][j] = sum_l f[l][c] P[l][j], P = softmax over the voxels l of the logits z, is GEMM shaped:
        #   df = P dmap^T  [L x M][M x C],   tt = F dmap  [L x C][C x M],   dz = P (tt - cj),  cj[j] = sum_c map[c][j] dmap[c][j]
        # — two plain library GEMMs in float32 and a few element-wise launches, for the LOWEST-RESOLUTION stage: k_mappool_bwd4
        # (one voxel per lane on the vector ALU) takes 435 us at 8^3, where its 8 workgroups leave the chip empty.  Measured
        # (MedFormer step, same box): form up to 8^3 33.57 -> 33.22 ms, up to 16^3 33.30, everywhere 36.1 (the softmax over
        # 32768 strided rows and the skinny float32 GEMMs cost more than the 195 us kernel).
        f3 = fw.reshape(N, Lr, Cf + M)
        P = torch.softmax(f3[..., Cf:].float(), dim=1)                      # [N, L, M]
        dm = dmap.float()
        G = torch.bmm(P, dm.transpose(1, 2))                                # [N, L, C]
        tt = torch.bmm(f3[..., :Cf].float(), dm)                            # [N, L, M]
        cj = (mp.float() * dm).sum(1, keepdim=True)                         # [N, 1, M]
        return torch.cat([G, P * (tt - cj)], -1).to(fw.dtype).reshape(fw.shape)
    dfw = torch.empty(tuple(fw.shape), dtype=fw.dtype, device=fw.device)
    check(_lib.lib().cbim_colsoftmax_pool_bwd(_dt(fw), _p(fw), _rs(fw), _p(mp), _p(cs), _p(dmap), _p(dfw), _rs(dfw), N, Lr,
                                              Cf, M, _stream(fw)), "colsoftmax_pool_bwd")
    return dfw


def trilinear_planes_fwd(x, size):
    """float32 [N,C,Di,Hi,Wi] -> [N,C,*size], align_corners=True."""
    _dev_ok(x)
    N, Cc, Di, Hi, Wi = map(int, x.shape)
    Do, Ho, Wo = (int(i) for i in size)
    y = torch.empty((N, Cc, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    check(_lib.lib().cbim_trilinear_planes_fwd(_p(x), _p(y), N * Cc, Di, Hi, Wi, Do, Ho, Wo, _stream(x)),
          "trilinear_planes_fwd")
    return y


def trilinear_planes_bwd(dy, in_shape):
    _dev_ok(dy)
    N, Cc, Di, Hi, Wi = map(int, in_shape)
    _, _, Do, Ho, Wo = map(int, dy.shape)
    if UP_SEPARABLE:   # W, H, D reductions of the NCDHW planes (scalar items along W, 16-byte chunks when the rows allow)
        kw = dict(dtype=torch.float32, device=dy.device)
        t = torch.empty((N, Cc, Do, Ho, Wi), **kw)
        _lin_adjoint(dy, 1, 0, t, N * Cc * Do * Ho, Wo, Wi, 1, vec=1)
        for (outer, F, L, inner, shape) in ((N * Cc * Do, Ho, Hi, Wi, (N, Cc, Do, Hi, Wi)), (N * Cc, Do, Di, Hi * Wi, (N, Cc, Di, Hi, Wi))):
            t2 = torch.empty(shape, **kw)
            _lin_adjoint(t, inner, 0, t2, outer, F, L, inner, vec=0 if inner % 4 == 0 else 1)
            t = t2
        return t
    dx = torch.empty(tuple(in_shape), dtype=torch.float32, device=dy.device)
    check(_lib.lib().cbim_trilinear_planes_bwd(_p(dy), _p(dx), N * Cc, Di, Hi, Wi, Do, Ho, Wo, _stream(dy)),
          "trilinear_planes_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# SwinUNETR shifted-window attention
# ------------------------------------------------------------------------------------------------

def _i3(v):
    return (C.c_int * 3)(*[int(i) for i in v])


def window_attn_fwd(qkv, qkv_bias, table, heads: int, window, shift, table_window):
    """qkv [B,D,H,W,3C]; table float32 [T,heads] -> out [B,D,H,W,C], lse float32 [num_windows, heads, 343]."""
    _dev_ok(qkv, qkv_bias, table)
    B, D, H, W, C3 = map(int, qkv.shape)
    Cc = C3 // 3
    L = _lib.lib()
    win = _i3(window)
    nwin = L.cbim_window_attn3d_num_windows(B, D, H, W, win)
    out = torch.empty((B, D, H, W, Cc), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((nwin, heads, 343), dtype=torch.float32, device=qkv.device)
    check(L.cbim_window_attn3d_fwd(_dt(qkv), _p(qkv), _p(qkv_bias), _p(table), _p(out), _p(lse), B, D, H, W, Cc, heads,
                                   win, _i3(shift), _i3(table_window), _stream(qkv)), "window_attn3d_fwd")
    return out, lse


def window_attn_bwd(qkv, qkv_bias, table, out, dout, lse, heads: int, window, shift, table_window):
    _dev_ok(qkv, qkv_bias, table, out, dout, lse)
    B, D, H, W, C3 = map(int, qkv.shape)
    Cc = C3 // 3
    L = _lib.lib()
    win, tw = _i3(window), _i3(table_window)
    nbytes = L.cbim_window_attn3d_workspace(B, D, H, W, Cc, heads, win, tw)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=qkv.device)
    dqkv = torch.empty_like(qkv)
    dtable = torch.empty(tuple(table.shape), dtype=torch.float32, device=qkv.device)
    dbias = torch.empty((C3,), dtype=torch.float32, device=qkv.device)
    check(L.cbim_window_attn3d_bwd(_dt(qkv), _p(qkv), _p(qkv_bias), _p(table), _p(out), _p(dout), _p(lse), _p(dqkv),
                                   _p(dtable), _p(dbias), B, D, H, W, Cc, heads, win, _i3(shift), tw, _p(ws), nbytes,
                                   _stream(qkv)), "window_attn3d_bwd")
    return dqkv, dtable, dbias


def layernorm_fwd(x, gamma, beta, eps: float, out_dtype: torch.dtype):
    """nn.LayerNorm over the last axis of x (float32 [..., C]) -> (y [..., C] in out_dtype, rowstats float32 [rows, 2])."""
    _dev_ok(x, gamma, beta)
    if x.dtype != torch.float32:
        raise TypeError("cbim_amd: layernorm_fwd takes the float32 residual stream")
    Cc = int(x.shape[-1])
    rows = x.numel() // Cc
    y = torch.empty(tuple(x.shape), dtype=out_dtype, device=x.device)
    rs = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().cbim_layernorm_fwd(_p(x), _p(gamma), _p(beta), float(eps), 1 if out_dtype == torch.bfloat16 else 0, _p(y),
                                        _p(rs), rows, Cc, _stream(x)), "layernorm_fwd")
    return y, rs


def layernorm_bwd(dy, x, gamma, rowstats, want_affine: bool, add=None):
    """-> (dx float32 [+ add], dgamma | None, dbeta | None)."""
    _dev_ok(dy, x, gamma, rowstats, add)
    if add is not None and (add.dtype != torch.float32 or add.numel() != x.numel()):
        raise TypeError("cbim_amd: layernorm_bwd `add` is the float32 gradient of the same rows")
    Cc = int(x.shape[-1])
    rows = x.numel() // Cc
    L = _lib.lib()
    dx = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    dg = db = ws = None
    nbytes = 0
    if want_affine:
        dg = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        db = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        nbytes = L.cbim_layernorm_bwd_workspace(rows, Cc)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    check(L.cbim_layernorm_bwd(_dt(dy), _p(dy), _p(x), _p(gamma), _p(rowstats), _p(add), _p(dx), _p(dg), _p(db), _p(ws), nbytes, rows, Cc,
                               _stream(x)), "layernorm_bwd")
    return dx, dg, db


_LIN_GEOM = {}


def linear_geom(Cin: int, Cout: int) -> ConvGeom:
    """the 1x1x1-convolution descriptor whose packed weight images the token Linear kernels read (bf16)"""
    g = _LIN_GEOM.get((Cin, Cout))
    if g is None:
        g = _LIN_GEOM[(Cin, Cout)] = ConvGeom(torch.bfloat16, 1, (1, 1, 1), Cin, Cout, (1, 1, 1), (0, 0, 0), 0)
    return g


_LO_CACHE = {}


def lo_weights(w: torch.Tensor, geom: ConvGeom, need_dgrad: bool):
    """(forward, dgrad | None) packed images of the rounding residue w - bf16(w) of a weight (cbim_conv3d_pack_weights_lo),
    re-packed when the parameter's version moved (once per optimizer step; under hipGraph capture the launch is captured)."""
    _dev_ok(w)
    key = (w.data_ptr(), tuple(w.shape))
    e = _LO_CACHE.get(key)
    if e is not None and e[3]() is not w:     # the address was recycled by another tensor (its finalizer has not run yet)
        e = None
    capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()      # (the pack launch must be IN a captured step)
    stale = PACKED.stale or capturing or e is None or e[0] != (w._version, PACKED.epoch) or (need_dgrad and e[2] is None)
    if stale:
        L = _lib.lib()
        p0 = e[1] if e is not None else torch.empty((L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), 0),), dtype=torch.uint8, device=w.device)
        p1 = e[2] if e is not None and e[2] is not None else (
            torch.empty((L.cbim_conv3d_packed_bytes(C.byref(geom.fwd), 1),), dtype=torch.uint8, device=w.device) if need_dgrad else None)
        check(L.cbim_conv3d_pack_weights_lo(C.byref(geom.fwd), _p(w.detach()), _p(p0), _p(p1), _stream(w)), "pack_weights_lo")
        if e is None:             # weakly keyed: the entry and its packed images go when the parameter does
            weakref.finalize(w, _LO_CACHE.pop, key, None)
        e = _LO_CACHE[key] = ((w._version, PACKED.epoch), p0, p1, weakref.ref(w))
    return e[1], (e[2] if need_dgrad else None)


def token_linear(x2d, w_packed, bias, Cout: int, act_in: int = 0, res=None, mask=None, mask_act: int = 0,
                 out_dtype: torch.dtype = torch.bfloat16, w_lo=None, out=None):
    """y = (act_in(x) @ W^T + bias) * act'(mask) + res over token rows (include/cbim_hip.h cbim_token_linear).
    x2d [rows, Cin] bf16 | fp32, res fp32 [rows, Cout] | None, mask bf16 [rows, Cout] | None -> y [rows, Cout] in out_dtype."""
    _rows_ok(x2d, w_packed, bias, res, mask, out)
    rows, Cin = int(x2d.shape[0]), int(x2d.shape[1])
    if res is not None and res.dtype != torch.float32:
        raise TypeError("cbim_amd: token_linear takes the float32 residual stream as `res`")
    if mask is not None and mask.dtype != torch.bfloat16:
        raise TypeError("cbim_amd: token_linear takes the bf16 pre-activation as `mask`")
    # out: a [rows, Cout] view with its own row stride (a channel slice of a wider row tensor) written in place
    y = torch.empty((rows, Cout), dtype=out_dtype, device=x2d.device) if out is None else out
    prof = PROFILE is not None and x2d.device.type == "cuda"
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().cbim_token_linear(_p(x2d), _dt(x2d), int(x2d.stride(0)), act_in, _p(w_packed), _p(w_lo), _p(bias),
                                       _p(res), int(res.stride(0)) if res is not None else 0,
                                       _p(mask), int(mask.stride(0)) if mask is not None else 0, mask_act,
                                       _p(y), _dt(y), int(y.stride(0)), rows, Cin, Cout, _stream(x2d)), "token_linear")
    if prof:
        e1.record()
        nbytes = rows * (Cin * x2d.element_size() + Cout * y.element_size() + (4 * Cout if res is not None else 0) +
                         (2 * Cout if mask is not None else 0)) + 2 * Cin * Cout
        PROFILE.append(("k_conv_pw<token>", 2.0 * rows * Cin * Cout, e0, e1, (Cin, Cout, rows, 1, 1), nbytes))
    return y


def token_linear_wgrad(x2d, dy2d, act_in: int = 0, out=None):
    """dW[co][ci] = sum_r dy[r][co] * act_in(x[r][ci]) -> float32 [Cout, Cin] (fixed summation order)."""
    _rows_ok(x2d, dy2d)
    rows, Cin, Cout = int(x2d.shape[0]), int(x2d.shape[1]), int(dy2d.shape[1])
    L = _lib.lib()
    nbytes = L.cbim_token_linear_wgrad_workspace(rows, Cin, Cout)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x2d.device)
    dw = _dw_out(out, (Cout, Cin), x2d.device)
    check(L.cbim_token_linear_wgrad(_p(x2d), _dt(x2d), int(x2d.stride(0)), act_in, _p(dy2d), _dt(dy2d), int(dy2d.stride(0)),
                                    _p(dw), _p(ws), nbytes, rows, Cin, Cout, _stream(x2d)), "token_linear_wgrad")
    return dw


def as_rows(t):
    """a gradient as the kernels can read it: the tensor itself when it is contiguous or a channel-slice view of a wider
    channels-last tensor (row stride > channel count: what torch.cat's backward hands out — no copy), else a contiguous copy"""
    return t if (t.is_contiguous() or _is_row_view(t)) else t.contiguous()


def colsum(x2d):
    """float32 [C] = column sums of x2d [rows, C] (bf16 or fp32), fixed summation order."""
    _dev_ok(x2d)
    rows, Cc = int(x2d.shape[0]), int(x2d.shape[1])
    L = _lib.lib()
    nbytes = L.cbim_colsum_workspace(rows, Cc)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x2d.device)
    out = torch.empty((Cc,), dtype=torch.float32, device=x2d.device)
    check(L.cbim_colsum(_dt(x2d), _p(x2d), rows, Cc, _p(out), _p(ws), nbytes, _stream(x2d)), "colsum")
    return out


def resnorm_fwd(a, stats_a, b, stats_b, act: int):
    """y = act(IN(a) + (IN(b) if stats_b is given else b))."""
    _dev_ok(a, stats_a, b, stats_b)
    N, Cc, S = int(a.shape[0]), int(a.shape[-1]), _spatial(a)
    y = torch.empty(tuple(a.shape), dtype=a.dtype, device=a.device)
    check(_lib.lib().cbim_resnorm_fwd(_dt(a), _p(a), _rs(a), _p(stats_a), _p(b), _rs(b), _p(stats_b), _p(y), Cc, N, S, Cc, act,
                                      _stream(a)), "resnorm_fwd")
    return y


def resnorm_bwd(dy, a, stats_a, b, stats_b, act: int, need_db: bool = True):
    _dev_ok(dy, a, stats_a, b, stats_b)
    N, Cc, S = int(a.shape[0]), int(a.shape[-1]), _spatial(a)
    L = _lib.lib()
    P = L.cbim_stats_parts(S, Cc)
    pa = torch.empty((N, P, Cc, 3), dtype=torch.float32, device=a.device)
    pb = torch.empty((N, P, Cc, 3), dtype=torch.float32, device=a.device) if stats_b is not None else None
    check(L.cbim_resnorm_bwd_reduce(_dt(a), _p(dy), _rs(dy), _p(a), _rs(a), _p(stats_a), _p(b), _rs(b), _p(stats_b), N, S, Cc, act,
                                    _p(pa), _p(pb), P, _stream(a)), "resnorm_bwd_reduce")
    ma = stats_finalize(pa, S, 0.0, 1)
    mb = stats_finalize(pb, S, 0.0, 1) if pb is not None else None
    da = torch.empty(tuple(a.shape), dtype=a.dtype, device=a.device)
    db = torch.empty(tuple(a.shape), dtype=a.dtype, device=a.device) if need_db else None
    check(L.cbim_resnorm_bwd_apply(_dt(a), _p(dy), _rs(dy), _p(a), _rs(a), _p(stats_a), _p(ma), _p(b), _rs(b), _p(stats_b),
                                   _p(mb), _p(da), _p(db), N, S, Cc, act, _stream(a)), "resnorm_bwd_apply")
    return da, db


def gate_fwd(x, psi):
    """y = x * psi; x [N,D,H,W,C], psi float32 [N,D,H,W] (one value per voxel)."""
    _dev_ok(x, psi)
    rows, Cc = x.numel() // int(x.shape[-1]), int(x.shape[-1])
    y = torch.empty_like(x)
    check(_lib.lib().cbim_gate_fwd(_dt(x), _p(x), _p(psi), _p(y), rows, Cc, _stream(x)), "gate_fwd")
    return y


def gate_bwd(dy, x, psi):
    _dev_ok(dy, x, psi)
    rows, Cc = x.numel() // int(x.shape[-1]), int(x.shape[-1])
    dx = torch.empty_like(x)
    dpsi = torch.empty(tuple(psi.shape), dtype=torch.float32, device=x.device)
    check(_lib.lib().cbim_gate_bwd(_dt(x), _p(dy), _p(x), _p(psi), _p(dx), _p(dpsi), rows, Cc, _stream(x)), "gate_bwd")
    return dx, dpsi
