"""ctypes binding of the C ABI in ``include/cbim_hip.h`` (libcbim_hip.so, gfx950).

The product library is ``libcbim_hip.so`` next to this file (built by
``__graft_entry__.build()`` / ``csrc/Makefile``).  There is NO fallback: if the library is
missing or an entry point fails, a RuntimeError is raised.  ``CBIM_HIP_LIBRARY`` may name an
explicit library file; the CPU test-suite uses it to load the host-side kernel executor built
from the very same sources under ``tests/emu`` (backend string "emu", CPU tensors only).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_lock = threading.Lock()
_lib = None

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "N", "Di", "Hi", "Wi", "Cin", "Do", "Ho", "Wo", "Cout",
        "kD", "kH", "kW", "pD", "pH", "pW", "act")]


_dp = C.POINTER(ConvDesc)

class PackItem(C.Structure):
    """cbim_pack_item of include/cbim_hip.h (one weight tensor of the one-launch weight re-layout)."""
    _fields_ = [("w0", C.c_void_p), ("w1", C.c_void_p), ("p0", C.c_void_p), ("p1", C.c_void_p),
                ("total0", C.c_int64), ("total1", C.c_int64)] + \
               [(n, C.c_int) for n in ("rows0", "Cout", "Cin", "taps", "BN0", "nch0", "BN1", "nch1", "block_begin",
                                       "n_blocks", "dtype", "_pad")]


class MapGemmDesc(C.Structure):
    """cbim_map_gemm_desc of include/cbim_hip.h (the small float32 GEMM of MedFormer's semantic-map branch)."""
    _fields_ = [("A", C.c_void_p), ("A2", C.c_void_p), ("lda", C.c_int64), ("a_batch", C.c_int64), ("a_t", C.c_int), ("a_split", C.c_int),
                ("X", C.c_void_p), ("X2", C.c_void_p), ("ldx", C.c_int64), ("x_batch", C.c_int64), ("x_t", C.c_int), ("x_split", C.c_int),
                ("OUT", C.c_void_p), ("OUT2", C.c_void_p), ("ldo", C.c_int64), ("o_batch", C.c_int64), ("o_t", C.c_int), ("o_split", C.c_int),
                ("R", C.c_void_p), ("ldr", C.c_int64), ("r_batch", C.c_int64),
                ("O", C.c_int), ("K", C.c_int), ("Nn", C.c_int), ("batch", C.c_int), ("reduce_batch", C.c_int), ("ln_mode", C.c_int),
                ("eps", C.c_float), ("_pad", C.c_int),
                ("Xn", C.c_void_p), ("rstd_out", C.c_void_p), ("XH", C.c_void_p), ("rstd_in", C.c_void_p)]


# name -> (restype, argtypes)   (mirrors include/cbim_hip.h one to one)
_SIGS = {
    "cbim_version": (i32, []),
    "cbim_backend": (C.c_char_p, []),
    "cbim_last_error_string": (C.c_char_p, []),
    "cbim_runtime_warmup": (i32, [vp]),
    "cbim_stats_parts": (i32, [i64, i32]),
    "cbim_instnorm_stats": (i32, [i32, vp, i64, i32, i64, i32, f32, vp, i32, vp, vp]),
    "cbim_stats_finalize": (i32, [vp, i32, i32, i32, f64, f32, i32, vp, vp]),
    "cbim_stats_restat": (i32, [vp, f32, f32, vp, i32, vp]),
    "cbim_se_fold_fwd": (i32, [vp, vp, f32, vp, vp, i32, vp]),
    "cbim_se_fold_bwd": (i32, [vp, vp, vp, f32, f64, vp, i32, vp]),
    "cbim_norm_act_fwd": (i32, [i32, vp, i64, vp, vp, i64, i32, i64, i32, i32, vp]),
    "cbim_norm_bwd_reduce": (i32, [i32, vp, i64, vp, i64, vp, i32, i64, i32, i32, i32, vp, i32, vp]),
    "cbim_norm_bwd_apply": (i32, [i32, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i32, i64, i32, i32, i32, vp]),
    "cbim_norm_affine_act_fwd": (i32, [i32, vp, i64, vp, vp, vp, i64, i32, i64, i32, i32, vp]),
    "cbim_norm_affine_bwd_reduce": (i32, [i32, vp, i64, vp, i64, vp, vp, i32, i64, i32, i32, i32, vp, i32, vp]),
    "cbim_norm_affine_bwd_apply": (i32, [i32, vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i64, i32, i32, i32, vp]),
    "cbim_bn_finish_fwd": (i32, [vp, i32, i32, C.c_double, f32, f32, vp, vp, i32, vp, vp, vp, vp, vp]),
    "cbim_bn_finish_bwd": (i32, [vp, i32, i32, C.c_double, vp, i32, vp, vp, vp, vp]),
    "cbim_maxpool3d_fwd": (i32, [i32, vp, vp, vp] + [i32] * 8 + [vp]),
    "cbim_maxpool3d_bwd": (i32, [i32, vp, vp, vp] + [i32] * 8 + [vp]),
    "cbim_upcat_fwd": (i32, [i32, vp, vp, vp] + [i32] * 10 + [vp]),
    "cbim_upcat_bwd": (i32, [i32, vp, vp, vp] + [i32] * 10 + [vp]),
    "cbim_upcat_fwd_stats": (i32, [i32, vp, vp, vp] + [i32] * 10 + [f32, vp, i32, vp, vp]),
    "cbim_up_stats": (i32, [i32, vp] + [i32] * 8 + [f32, vp, i32, vp, vp]),
    "cbim_up_gram_parts": (i32, [i32, i32, i32]),
    "cbim_up_stats_gram": (i32, [i32, vp] + [i32] * 8 + [f32, vp, i32, vp, vp]),
    "cbim_upcat_act_fwd": (i32, [i32, vp, vp, vp, vp] + [i32] * 11 + [vp]),
    "cbim_upcat_norm_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp] + [i32] * 10 + [vp]),
    "cbim_up_tile_parts": (i32, [i32, i32, i32]),
    "cbim_up_tile_min_tiles": (i64, [i64]),
    "cbim_up_stats_tile": (i32, [i32, vp] + [i32] * 8 + [f32, vp, i32, vp, vp]),
    "cbim_upcat_act_fwd_tile": (i32, [i32, vp, vp, vp, vp] + [i32] * 11 + [vp]),
    "cbim_upcat_norm_bwd_tile": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp] + [i32] * 10 + [vp]),
    "cbim_lin_adjoint_axis": (i32, [i32, i32, vp, i64, i64, vp, i64, i32, i32, i64, vp]),
    "cbim_conv3d_packed_bytes": (sz, [_dp, i32]),
    "cbim_conv3d_pack_weights": (i32, [_dp, i32, vp, vp, vp]),
    "cbim_conv3d_pack_weights_both": (i32, [_dp, vp, vp, vp, vp]),
    "cbim_conv3d_pack_weights_lo": (i32, [_dp, vp, vp, vp, vp]),
    "cbim_conv3d_pack_item_fill": (i32, [_dp, vp, vp, i32, vp, vp, i32, vp]),
    "cbim_conv3d_pack_weights_table": (i32, [vp, i32, i32, i32, vp]),
    "cbim_conv3d_last_kernel": (i32, []),
    "cbim_conv_r32_min_voxels": (i64, [i64]),
    "cbim_conv_r32_tile_depth": (i32, [i32]),
    "cbim_conv_rw_enable": (i32, [i32, i32]),
    "cbim_conv_rw48_enable": (i32, [i32]),
    "cbim_conv_rw48_takes": (i32, [_dp]),
    "cbim_conv_pw_enable": (i32, [i32]),
    "cbim_dwconv_lds_enable": (i32, [i32]),
    "cbim_conv3d_num_tiles": (i32, [_dp]),
    "cbim_conv3d_tile_config": (i32, [_dp, C.POINTER(C.c_int * 4)]),
    "cbim_conv3d_igemm": (i32, [_dp, vp, i64, vp, i64, i32, vp, vp, vp, i64, vp, i64, vp, vp, i64, vp, vp, sz, vp]),
    "cbim_conv3d_igemm_workspace": (sz, [_dp]),
    "cbim_conv3d_wgrad_workspace": (sz, [_dp]),
    "cbim_conv3d_wgrad_last_kernel": (i32, []),
    "cbim_wgrad_r32_enable": (i32, [i32]),
    "cbim_wgrad_r32_waves": (i32, [i32]),
    "cbim_conv3d_wgrad": (i32, [_dp, vp, i64, vp, i64, i32, vp, vp, i64, vp, i64, i32, vp, vp, sz, vp]),
    "cbim_stem_conv_fwd": (i32, [i32, vp, vp, vp] + [i32] * 15 + [vp]),
    "cbim_stem_conv_wgrad_workspace": (sz, [i32] * 9),
    "cbim_stem_conv_wgrad": (i32, [i32, vp, vp, vp] + [i32] * 15 + [vp, sz, vp]),
    "cbim_head_fwd": (i32, [i32, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "cbim_head_bwd_workspace": (sz, [i64, i32, i32, i32]),
    "cbim_head_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, sz, vp]),
    "cbim_head_mfma_enable": (i32, [i32]),
    "cbim_stem_mfma_enable": (i32, [i32]),
    "cbim_dice_ce_workspace": (sz, [i32, i32, i64]),
    "cbim_dice_ce_fwd": (i32, [vp, vp, vp, i32, i32, i64, vp, vp, vp, sz, vp]),
    "cbim_dice_ce_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i64, vp]),
    "cbim_affine_sample3d": (i32, [vp, vp, i32, vp, vp, vp] + [i32] * 10 + [vp]),
    "cbim_crop3d": (i32, [vp, vp, i32, vp, vp] + [i32] * 10 + [vp]),
    "cbim_chan_stats_workspace": (sz, [i32, i64]),
    "cbim_chan_stats": (i32, [vp, i32, i64, vp, vp, sz, vp]),
    "cbim_intensity": (i32, [vp, vp, i32, i64, i32, vp, i32, vp, vp, i32, vp, vp]),
    "cbim_gaussian_blur3d": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "cbim_dwconv3d": (i32, [i32, vp, i64, vp, i32, vp, vp, i32, vp, i64] + [i32] * 8 + [vp]),
    "cbim_dwconv3d_wgrad_workspace": (sz, [i32] * 8),
    "cbim_dwconv3d_wgrad": (i32, [i32, vp, i64, vp, i32, vp, i64, vp, vp] + [i32] * 8 + [vp, sz, vp]),
    "cbim_space_to_depth": (i32, [i32, vp, vp] + [i32] * 9 + [vp]),
    "cbim_space_to_depth_strided": (i32, [i32, vp, vp] + [i32] * 9 + [i64, vp]),
    "cbim_attn_wide_max_codes": (i32, []),
    "cbim_bidir_attn_workspace": (sz, [i32] * 5),
    "cbim_bidir_attn_fwd": (i32, [i32, vp, i64, vp, vp, vp, vp, vp] + [i32] * 5 + [f32, vp, sz, vp]),
    "cbim_bidir_attn_bwd": (i32, [i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp] + [i32] * 5 + [f32, vp, sz, vp]),
    "cbim_map_gemm": (i32, [C.POINTER(MapGemmDesc), vp]),
    "cbim_se_gate_fwd": (i32, [vp] * 7 + [i32] * 3 + [vp]),
    "cbim_se_gate_bwd": (i32, [vp] * 12 + [i32] * 3 + [vp]),
    "cbim_awg_rows": (i32, [vp, i64, i32, i32, f32, vp, vp, vp]),
    "cbim_awg_cols": (i32, [vp, i64, i32, i32, f32, vp, vp, vp, vp]),
    "cbim_awg_ds": (i32, [vp] * 6 + [i32, i64, i32, i32, f32, vp, vp, vp]),
    "cbim_colsoftmax_pool_workspace": (sz, [i32] * 4),
    "cbim_colsoftmax_pool_fwd": (i32, [i32, vp, i64, vp, vp] + [i32] * 4 + [vp, sz, vp]),
    "cbim_colsoftmax_pool_bwd": (i32, [i32, vp, i64, vp, vp, vp, vp, i64] + [i32] * 4 + [vp]),
    "cbim_trilinear_planes_fwd": (i32, [vp, vp] + [i32] * 7 + [vp]),
    "cbim_trilinear_planes_bwd": (i32, [vp, vp] + [i32] * 7 + [vp]),
    "cbim_resnorm_fwd": (i32, [i32, vp, i64, vp, vp, i64, vp, vp, i64, i32, i64, i32, i32, vp]),
    "cbim_resnorm_bwd_reduce": (i32, [i32, vp, i64, vp, i64, vp, vp, i64, vp, i32, i64, i32, i32, vp, vp, i32, vp]),
    "cbim_resnorm_bwd_apply": (i32, [i32, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "cbim_window_attn3d_num_windows": (i32, [i32] * 4 + [vp]),
    "cbim_window_attn3d_workspace": (sz, [i32] * 6 + [vp, vp]),
    "cbim_window_attn3d_fwd": (i32, [i32, vp, vp, vp, vp, vp] + [i32] * 6 + [vp, vp, vp, vp]),
    "cbim_window_attn3d_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp] + [i32] * 6 + [vp, vp, vp, vp, sz, vp]),
    "cbim_optim_chunk": (i32, []),
    "cbim_adamw_ema_step": (i32, [vp, vp, vp, i32, vp, vp]),
    "cbim_ema_step": (i32, [vp, vp, vp, i32, f32, f32, vp]),
    "cbim_softmax_accumulate": (i32, [vp, vp, vp] + [i32] * 11 + [vp]),
    "cbim_prob_finalize": (i32, [vp, vp, vp, i32, i32, i64, vp]),
    "cbim_dice_counts": (i32, [vp, i32, vp, i32, i64, i64, i32, vp, vp]),
    "cbim_gate_fwd": (i32, [i32, vp, vp, vp, i64, i32, vp]),
    "cbim_gate_bwd": (i32, [i32, vp, vp, vp, vp, vp, i64, i32, vp]),
    "cbim_layernorm_fwd": (i32, [vp, vp, vp, f32, i32, vp, vp, i64, i32, vp]),
    "cbim_layernorm_bwd_workspace": (sz, [i64, i32]),
    "cbim_layernorm_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, i32, vp]),
    "cbim_token_linear": (i32, [vp, i32, i64, i32, vp, vp, vp, vp, i64, vp, i64, i32, vp, i32, i64, i64, i32, i32, vp]),
    "cbim_token_linear_wgrad_workspace": (sz, [i64, i32, i32]),
    "cbim_token_linear_wgrad": (i32, [vp, i32, i64, i32, vp, i32, i64, vp, vp, sz, i64, i32, i32, vp]),
    "cbim_colsum_workspace": (sz, [i64, i32]),
    "cbim_colsum": (i32, [i32, vp, i64, i32, vp, vp, sz, vp]),
    "cbim_ncdhw_to_ndhwc": (i32, [i32, vp, vp, i32, i32, i64, vp]),
    "cbim_ndhwc_to_ncdhw": (i32, [i32, vp, vp, i32, i32, i64, vp]),
}

EXPORTS = tuple(_SIGS)


def library_path() -> str:
    return os.environ.get("CBIM_HIP_LIBRARY") or os.path.join(_HERE, "libcbim_hip.so")


def lib():
    """Load (once) and return the bound library; raises if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.isfile(path):
            raise RuntimeError(
                f"cbim_amd: HIP kernel library not found at {path}; build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (there is no fallback path)")
        # PyTorch's ROCm wheel bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
        # libcbim_hip.so must bind to THAT instance — device pointers and streams are torch's — which the dynamic
        # loader does by SONAME only if torch's libraries are already mapped.  Loaded the other way round, this
        # library pulls /opt/rocm's runtime as a second instance and every launch fails with
        # "no ROCm-capable device is detected".
        import torch  # noqa: F401
        h = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(h, name)
            except AttributeError as e:  # pragma: no cover
                raise RuntimeError(f"cbim_amd: {path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = h
        return _lib


def backend() -> str:
    return lib().cbim_backend().decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().cbim_last_error_string().decode()
        raise RuntimeError(f"cbim_amd: {what} failed with code {rc}: {msg}")
