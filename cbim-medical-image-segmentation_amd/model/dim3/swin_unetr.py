"""SwinUNETR behind the reference's constructor signature and parameter names
(/root/reference/model/dim3/swin_unetr.py:32-292 and the monai 1.1.0 blocks it imports, :24-27).

Execution split:
  * conv encoder/decoder (92 % of the FLOPs: UnetrBasicBlock / UnetrUpBlock / UnetOutBlock) — the engine's conv kernels: the
    48-channel layers of the 128^3 / 64^3 levels on k_conv3_rw48 / k_wgrad_r32 with act(IN(x)) materialised once (round 6),
    the others with InstanceNorm(eps 1e-5) + LeakyReLU fused on load; statistics in the epilogues, the post-norm residual tail as
    one streaming kernel (``functional.ResNormFn``), ConvTranspose3d(k=2,s=2) as a 1x1 GEMM + depth-to-space scatter;
  * shifted-window attention — one kernel per block does pad/roll/partition/bias/mask/softmax/AV/reverse/crop
    (``functional.WindowAttnFn``);
  * the token-wise Linear layers (qkv, proj, MLP, patch merging/embedding) on the engine's row GEMM (``functional.TokenLinearFn``:
    bias, GELU on load, GELU' mask, fp32 residual add in the kernel) in the bf16 engine mode — the fp32 parity mode keeps torch's
    fp32 GEMMs unless ``set_fp32_token_gemm(True)``; LayerNorm is ``k_layernorm_fwd/bwd``.  The residual stream and LayerNorm
    of the trunk stay fp32; in the bf16 engine mode the Linears, GELU and the window-attention kernel of each block run in bf16
    (as the reference under AMP).
forward(x[B,C,D,H,W] fp32 NCDHW) -> logits[B,classes,D,H,W] fp32.
"""
import itertools

import torch
import torch.nn as nn

import torch.nn.functional as F

from ... import functional as Fn
from ...functional import eager_only
from ...ops import ACT

from ...ops import ncdhw_to_ndhwc as ops_ncdhw_to_ndhwc

_EPS = 1e-5          # nn.InstanceNorm3d default (monai get_norm_layer("instance"))
_LRELU = ACT["lrelu"]  # monai UnetResBlock act: LeakyReLU(0.01)
_GELU = ACT["gelu"]    # monai MLPBlock act


def _tup3(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * 3


# ---- monai conv blocks ---------------------------------------------------------------------------------

class _ConvHolder(nn.Module):
    """monai ``Convolution(conv_only=True)``: a module whose single child ``conv`` owns the weight."""

    def __init__(self, cin, cout, k, bias=False, transposed=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=k, bias=bias)
        else:
            self.conv = nn.Conv3d(cin, cout, kernel_size=k, stride=1, padding=(k - 1) // 2, bias=bias)


class UnetResBlock(nn.Module):
    """conv3-IN-lrelu-conv3-IN, residual (1x1 conv + IN when the channel count changes), add, lrelu."""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__()
        self.conv1 = _ConvHolder(in_channels, out_channels, kernel_size)
        self.conv2 = _ConvHolder(out_channels, out_channels, kernel_size)
        self.lrelu = nn.Identity()
        self.norm1 = nn.Identity()   # InstanceNorm3d(affine=False): fused into the next conv's load
        self.norm2 = nn.Identity()   # fused into the residual tail
        self.downsample = in_channels != out_channels
        if self.downsample:
            self.conv3 = _ConvHolder(in_channels, out_channels, 1)
            self.norm3 = nn.Identity()

    def forward(self, x, stem_dtype=None):
        """x: channels-last feature map, or (stem_dtype given) the NCDHW fp32 network input."""
        if stem_dtype is not None and stem_dtype == torch.bfloat16 and x.shape[1] <= 8:
            # the network input as an 8-channel (zero-padded) channels-last bf16 tensor: conv1 / conv3 of encoder1 then run
            # on the matrix cores like every other conv (k_stem_fwd / k_stem_wgrad<4>, direct vector-ALU kernels, were
            # 0.9 + 1.8 ms of the 32 ms step); the padded weight columns are zero and their gradient is sliced off
            cin = int(x.shape[1])
            xp = F.pad(x, (0, 0, 0, 0, 0, 0, 0, 8 - cin)) if cin < 8 else x
            x = ops_ncdhw_to_ndhwc(xp.contiguous(), stem_dtype)
            w1 = F.pad(self.conv1.conv.weight, (0, 0, 0, 0, 0, 0, 0, 8 - cin))
            z1, s1 = Fn.NormConvFn.apply(x, None, w1, 0, None, True, None, _EPS)
            z2, s2 = Fn.NormConvFn.apply(z1, s1, self.conv2.conv.weight, _LRELU, None, True, None, _EPS)
            if not self.downsample:
                raise NotImplementedError("cbim_amd: identity-residual UnetResBlock on the raw network input is not built")
            w3 = F.pad(self.conv3.conv.weight, (0, 0, 0, 0, 0, 0, 0, 8 - cin))
            r, s3 = Fn.NormConvFn.apply(x, None, w3, 0, None, True, None, _EPS)
            return Fn.ResNormFn.apply(z2, s2, r, s3, _LRELU)
        if stem_dtype is not None:
            z1 = Fn.StemFn.apply(x, self.conv1.conv.weight, stem_dtype)
            s1 = Fn.ensure_stats(Fn.FMap(z1, None), _EPS).stats
        elif self.downsample:
            # conv1 and the 1x1x1 residual conv3 read the same tensor: one autograd node, the two input gradients accumulated in
            # the kernel (functional.DualRawConvFn)
            z1, s1, r, s3 = Fn.DualRawConvFn.apply(x, self.conv1.conv.weight, self.conv3.conv.weight, _EPS)
            z2, s2 = Fn.NormConvFn.apply(z1, s1, self.conv2.conv.weight, _LRELU, None, True, None, _EPS)
            return Fn.ResNormFn.apply(z2, s2, r, s3, _LRELU)
        else:
            z1, s1 = Fn.NormConvFn.apply(x, None, self.conv1.conv.weight, 0, None, True, None, _EPS)
        z2, s2 = Fn.NormConvFn.apply(z1, s1, self.conv2.conv.weight, _LRELU, None, True, None, _EPS)
        if self.downsample:
            if stem_dtype is not None:
                r = Fn.StemFn.apply(x, self.conv3.conv.weight, stem_dtype)
                s3 = Fn.ensure_stats(Fn.FMap(r, None), _EPS).stats
            else:
                r, s3 = Fn.NormConvFn.apply(x, None, self.conv3.conv.weight, 0, None, True, None, _EPS)
            return Fn.ResNormFn.apply(z2, s2, r, s3, _LRELU)
        if stem_dtype is not None:
            raise NotImplementedError("cbim_amd: identity-residual UnetResBlock on the raw network input is not built")
        return Fn.ResNormFn.apply(z2, s2, x, None, _LRELU)


class UnetrBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
        super().__init__()
        if spatial_dims != 3 or stride != 1 or norm_name != "instance" or not res_block:
            raise NotImplementedError("cbim_amd: only UnetrBasicBlock(3-D, stride 1, instance norm, res_block) is built")
        self.layer = UnetResBlock(in_channels, out_channels, kernel_size)

    def forward(self, x, stem_dtype=None):
        return self.layer(x, stem_dtype)


class UnetrUpBlock(nn.Module):
    """ConvTranspose3d(k=2,s=2,bias=False) -> cat([up, skip]) -> UnetResBlock."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name,
                 res_block=False):
        super().__init__()
        if spatial_dims != 3 or upsample_kernel_size != 2 or norm_name != "instance" or not res_block:
            raise NotImplementedError("cbim_amd: only UnetrUpBlock(3-D, upsample 2, instance norm, res_block) is built")
        self.transp_conv = _ConvHolder(in_channels, out_channels, 2, transposed=True)
        self.conv_block = UnetResBlock(out_channels + out_channels, out_channels, kernel_size)

    def forward(self, x, skip):
        w = self.transp_conv.conv.weight                               # [Cin, Cout, 2, 2, 2]
        cout = w.shape[1]
        w_eq = w.permute(2, 3, 4, 1, 0).reshape(8 * cout, w.shape[0], 1, 1, 1)
        t, _ = Fn.NormConvFn.apply(x, None, w_eq, 0, None, False, None, _EPS)
        if skip.is_contiguous() and skip.dtype == t.dtype and tuple(skip.shape[1:4]) == tuple(2 * v for v in t.shape[1:4]):
            return self.conv_block(Fn.UpCatSkipFn.apply(t, skip, (2, 2, 2)))     # round 6: the up-sampled tensor is never stored
        up = Fn.DepthToSpaceFn.apply(t, (2, 2, 2))
        return self.conv_block(torch.cat((up, skip), dim=-1))


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels):
        super().__init__()
        self.conv = _ConvHolder(in_channels, out_channels, 1, bias=True)


# ---- Swin transformer trunk ----------------------------------------------------------------------------

class PatchEmbed(nn.Module):
    """monai PatchEmbed: Conv3d(k = stride = patch) as one GEMM over the 2x2x2xC patch."""

    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = _tup3(patch_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)

    def forward(self, x):
        B, Cc, D, H, W = x.shape
        p = self.patch_size
        if D % p[0] or H % p[1] or W % p[2]:
            x = F.pad(x, (0, (-W) % p[2], 0, (-H) % p[1], 0, (-D) % p[0]))
            _, _, D, H, W = x.shape
        x = x.reshape(B, Cc, D // p[0], p[0], H // p[1], p[1], W // p[2], p[2]).permute(0, 2, 4, 6, 1, 3, 5, 7)
        tok = x.reshape(B, D // p[0], H // p[1], W // p[2], -1)
        if _in_tree(tok, self.proj) or _tree32(tok, self.proj):
            # the fp32 patch rows feed the row GEMM as they are (bf16 hi + lo fragments of rows AND weights: this Linear lies
            # outside the blocks' reduced-precision region and keeps fp32 accuracy); fp32 out = the residual stream
            return Fn.token_linear(tok.contiguous(), self.proj.weight, self.proj.bias, out_dtype=torch.float32, need_dx=False,
                                   exact=True)
        return F.linear(tok, self.proj.weight.flatten(1), self.proj.bias)      # (the fp32 parity mode: torch's fp32 GEMM)


# The token Linears of the trunk run on the engine's own row-GEMM kernel in the bf16 engine mode (functional.TokenLinearFn: bias,
# GELU on load, GELU' mask and the fp32 residual add in the kernel; round 5); the fp32 parity mode keeps torch's fp32 GEMMs unless
# set_fp32_token_gemm(True) (round 6, below).


def _on_engine_device(x) -> bool:
    from ... import _lib
    return x.device.type == ("cpu" if _lib.backend() == "emu" else "cuda")


def _trunk_bf16(x) -> bool:
    """True when the trunk's token Linears / GELU / window attention run in bf16 (the bf16 engine mode, like the reference under
    AMP): LayerNorm and the residual stream stay fp32"""
    return _on_engine_device(x) and Fn.compute_dtype() == torch.bfloat16


def _in_tree(x, lin) -> bool:
    w = lin.weight
    cin = w.numel() // w.shape[0]
    return (_trunk_bf16(x) and x.dtype in (torch.bfloat16, torch.float32) and cin % 8 == 0 and w.shape[0] % 8 == 0
            and cin <= 4096 and w.dtype == torch.float32)


# round 6: the fp32 PARITY mode can also run the trunk's Linears on the engine's row GEMM — fp32 rows as bf16 hi + lo fragments
# against the weight image AND its residue image (three MFMAs per fragment pair: an fp32-accurate product), fp32 out — so that the
# fp32 goldens pin k_conv_pw's token mode at model level instead of hipBLASLt.  Off by default: the weight
# gradient of that kernel rounds its operands to bf16, so the default fp32 mode keeps torch's fp32 GEMMs for the backward's sake.
_FP32_TOKEN_GEMM = False


def set_fp32_token_gemm(on: bool) -> bool:
    global _FP32_TOKEN_GEMM
    old, _FP32_TOKEN_GEMM = _FP32_TOKEN_GEMM, bool(on)
    return old


def _tree32(x, lin) -> bool:
    w = lin.weight
    cin = w.numel() // w.shape[0]
    return (_FP32_TOKEN_GEMM and _on_engine_device(x) and Fn.compute_dtype() == torch.float32 and x.dtype == torch.float32
            and cin % 8 == 0 and w.shape[0] % 8 == 0 and cin <= 4096 and w.dtype == torch.float32)


def _lin32(lin, x, res=None):
    return Fn.token_linear(x, lin.weight, lin.bias, res=res, out_dtype=torch.float32, exact=True)


def _token_linear(lin, x):
    """the fp32 parity mode's Linear: torch's own GEMM (and its autograd)"""
    return lin(x)




def _layer_norm(ln, x, out_dtype):
    """nn.LayerNorm `ln` applied by the HIP kernel (fp32 rows in, `out_dtype` out) — torch's own op for shapes it does not take"""
    if x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 3072:
        return Fn.layer_norm(x, ln.weight, ln.bias, ln.eps, out_dtype)
    return ln(x)


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)

    def forward(self, x, res=None):
        """res: the fp32 residual stream to add (the block's `x + mlp(norm2(x))`, swin_unetr.py:552,640-643)"""
        if _tree32(x, self.linear1) and _tree32(x, self.linear2):
            return _lin32(self.linear2, F.gelu(_lin32(self.linear1, x)), res)
        if _in_tree(x, self.linear1) and _in_tree(x, self.linear2):
            h = Fn.token_linear(x, self.linear1.weight, self.linear1.bias)                       # pre-activation, bf16
            return Fn.token_linear(h, self.linear2.weight, self.linear2.bias, act_in=_GELU, res=res,   # GELU on load, + res
                                   out_dtype=torch.float32 if res is not None else torch.bfloat16)
        y = _token_linear(self.linear2, F.gelu(_token_linear(self.linear1, x)))
        return y if res is None else res + y


def _relative_position_index(ws):
    coords = torch.stack(torch.meshgrid(*[torch.arange(w) for w in ws], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    for i in range(3):
        rel[:, :, i] += ws[i] - 1
    rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
    rel[:, :, 1] *= 2 * ws[2] - 1
    return rel.sum(-1)


class WindowAttention(nn.Module):
    """swin_unetr.py:384-490.  ``relative_position_index`` is kept as a buffer for state_dict compatibility; the
    kernel derives the same index arithmetically."""

    def __init__(self, dim, num_heads, window_size, qkv_bias=False):
        super().__init__()
        self.dim, self.num_heads, self.window_size = dim, num_heads, _tup3(window_size)
        ws = self.window_size
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1), num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, h, window, shift, res=None):
        """res: the fp32 residual stream to add (the block's `x + attn(norm1(x))`, swin_unetr.py:539-549)"""
        tree = _in_tree(h, self.qkv) and _in_tree(h, self.proj)
        if _tree32(h, self.qkv) and _tree32(h, self.proj):
            o = Fn.WindowAttnFn.apply(_lin32(self.qkv, h), self.qkv.bias, self.relative_position_bias_table, self.num_heads, window, shift,
                                      self.window_size)
            return _lin32(self.proj, o, res)
        qkv = Fn.token_linear(h, self.qkv.weight, self.qkv.bias) if tree else _token_linear(self.qkv, h)
        o = Fn.WindowAttnFn.apply(qkv, self.qkv.bias, self.relative_position_bias_table, self.num_heads, window, shift,
                                  self.window_size)
        if tree:
            return Fn.token_linear(o, self.proj.weight, self.proj.bias, res=res,
                                   out_dtype=torch.float32 if res is not None else torch.bfloat16)
        y = _token_linear(self.proj, o)
        return y if res is None else res + y


class SwinTransformerBlock(nn.Module):
    """swin_unetr.py:493-656."""

    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.window_size, self.shift_size = _tup3(window_size), _tup3(shift_size)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, num_heads, self.window_size, qkv_bias=qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLPBlock(dim, int(dim * mlp_ratio))

    # checkpoint name -> attribute path here (the checkpoint's MLP is `fc1` / `fc2`, monai's MLPBlock calls them linear1 / linear2)
    _CKPT = (("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"),
             ("attn.relative_position_bias_table", "attn.relative_position_bias_table"),
             ("attn.relative_position_index", "attn.relative_position_index"),
             ("attn.qkv.weight", "attn.qkv.weight"), ("attn.qkv.bias", "attn.qkv.bias"),
             ("attn.proj.weight", "attn.proj.weight"), ("attn.proj.bias", "attn.proj.bias"),
             ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"),
             ("mlp.fc1.weight", "mlp.linear1.weight"), ("mlp.fc1.bias", "mlp.linear1.bias"),
             ("mlp.fc2.weight", "mlp.linear2.weight"), ("mlp.fc2.bias", "mlp.linear2.bias"))

    def load_from(self, weights, n_block, layer):
        """swin_unetr.py:610-643: the fourteen tensors of block `n_block` of `layer` from a Swin-ViT checkpoint"""
        root = f"module.{layer}.0.blocks.{n_block}."
        with torch.no_grad():
            for src, dst in self._CKPT:
                t = self
                for part in dst.split("."):
                    t = getattr(t, part)
                t.copy_(weights["state_dict"][root + src])

    def forward(self, x):
        dims = tuple(x.shape[1:4])
        ws = tuple(d if d <= w else w for d, w in zip(dims, self.window_size))            # get_window_size (:358-381)
        ss = tuple(0 if d <= w else s for d, w, s in zip(dims, self.window_size, self.shift_size))
        # bf16 engine mode: the token Linears (qkv, proj, MLP), GELU and the window-attention kernel run in bf16 like
        # the reference under AMP (LayerNorm and the residual stream stay fp32); fp32 mode is untouched
        amp = _trunk_bf16(x)
        nd = torch.bfloat16 if amp else torch.float32       # the LayerNorm kernel stores what the Linears consume
        with torch.autocast(x.device.type, dtype=torch.bfloat16, enabled=amp):      # (the Linears that stay on torch: odd widths)
            if x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 3072:
                # (LN(x), x) as one autograd node: its backward adds the residual-stream gradient inside the LayerNorm kernel
                y, xr = Fn.LayerNormResFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, nd)
                x = self.attn(y, ws, ss, res=xr)
                y, xr = Fn.LayerNormResFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, nd)
                return self.mlp(y, res=xr)
            x = self.attn(_layer_norm(self.norm1, x, nd), ws, ss, res=x)
            return self.mlp(_layer_norm(self.norm2, x, nd), res=x)


class PatchMerging(nn.Module):
    """The v0.9 ``PatchMerging`` (swin_unetr.py:707-731): note the slice list repeats three octants."""
    _SEL = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1))

    _SEL_CACHE = {}

    @classmethod
    def _sel_index(cls, device):
        """octant numbers (4 i + 2 j + k) of the slice list, resident on `device` (created once: a host -> device copy inside
        a hipGraph capture is not allowed)"""
        t = cls._SEL_CACHE.get(str(device))
        if t is None:
            t = torch.tensor([4 * i + 2 * j + k for i, j, k in cls._SEL], device=device)
            cls._SEL_CACHE[str(device)] = t
        return t

    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim)

    def forward(self, x):
        _, d, h, w, _ = x.shape
        if (d % 2) or (h % 2) or (w % 2):
            x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2, 0, d % 2))
        cdiv = 8 if x.dtype == torch.bfloat16 else 4
        if x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % cdiv == 0:
            # the 8 strided slices as ONE gather kernel (octant o = 4 i + 2 j + k), then the v0.9 slice order — with its
            # repeated octants — as an index_select over the 8 channel blocks.  The slice-by-slice form costs autograd one
            # zero-filled full-size tensor and one strided add per slice in the backward (16 launches per layer).
            B, d2, h2, w2, Cc = x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3] // 2, x.shape[4]
            m = Fn.SpaceToDepthFn.apply(x.contiguous(), (2, 2, 2)).view(B, d2, h2, w2, 8, Cc)
            x = m.index_select(4, self._sel_index(x.device)).reshape(B, d2, h2, w2, 8 * Cc)
        else:
            x = torch.cat([x[:, i::2, j::2, k::2, :] for i, j, k in self._SEL], -1)
        if _in_tree(x, self.reduction) or _tree32(x, self.reduction):
            # fp32 LayerNorm rows feed the row GEMM as they are (bf16 hi + lo fragments); fp32 out = the next stage's stream
            return Fn.token_linear(_layer_norm(self.norm, x, torch.float32), self.reduction.weight, None, out_dtype=torch.float32,
                                   exact=True)
        return _token_linear(self.reduction, _layer_norm(self.norm, x, torch.float32))


class BasicLayer(nn.Module):
    """swin_unetr.py:776-873 (channels-last throughout; the reference's NCDHW<->NDHWC rearranges disappear)."""

    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=False):
        super().__init__()
        ws = _tup3(window_size)
        shift = tuple(i // 2 for i in ws)
        self.blocks = nn.ModuleList([SwinTransformerBlock(dim, num_heads, ws, (0, 0, 0) if i % 2 == 0 else shift,
                                                          mlp_ratio=mlp_ratio, qkv_bias=qkv_bias) for i in range(depth)])
        self.downsample = PatchMerging(dim)

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x)


class SwinTransformer(nn.Module):
    """swin_unetr.py:876-997; returns the five channel-LayerNormed hidden states, channels-last."""

    def __init__(self, in_chans, embed_dim, window_size, patch_size, depths, num_heads, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        self.pos_drop = nn.Identity()
        for i, name in enumerate(("layers1", "layers2", "layers3", "layers4")):
            setattr(self, name, nn.ModuleList([BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size,
                                                          mlp_ratio=mlp_ratio, qkv_bias=qkv_bias)]))

    def forward(self, x, normalize=True):
        od = Fn.compute_dtype()     # the conv encoder reads the hidden states in the engine's activation dtype

        def out(t):   # proj_out (:970-983): F.layer_norm(x, [ch]) without affine parameters
            if not normalize:
                return t
            if t.dtype == torch.float32 and t.shape[-1] % 4 == 0 and t.shape[-1] <= 3072:
                return Fn.layer_norm(t, None, None, 1e-5, od)
            return F.layer_norm(t, (t.shape[-1],))
        x = self.patch_embed(x)
        outs = [out(x)]
        for layers in (self.layers1, self.layers2, self.layers3, self.layers4):
            x = layers[0](x)
            outs.append(out(x))
        return outs


class SwinUNETR(nn.Module):
    def __init__(self, img_size, in_channels, out_channels, depths=(2, 2, 2, 0), num_heads=(3, 6, 12, 24), feature_size=24,
                 norm_name="instance", drop_rate=0.0, attn_drop_rate=0.0, dropout_path_rate=0.0, normalize=True,
                 use_checkpoint=False, spatial_dims=3, downsample="merging"):
        super().__init__()
        if spatial_dims != 3 or downsample != "merging" or drop_rate or attn_drop_rate or dropout_path_rate or use_checkpoint:
            raise NotImplementedError("cbim_amd: SwinUNETR is built for 3-D, 'merging' downsampling, no dropout / checkpointing")
        img_size = _tup3(img_size)
        for m in img_size:                       # swin_unetr.py:96-99
            if m % 32 != 0:
                raise ValueError("input image size (img_size) should be divisible by stage-wise image resolution.")
        if feature_size % 12 != 0:
            raise ValueError("feature_size should be divisible by 12.")
        self.normalize = normalize
        f = feature_size
        self.swinViT = SwinTransformer(in_channels, f, (7, 7, 7), (2, 2, 2), depths, num_heads, mlp_ratio=4.0, qkv_bias=True)
        kw = dict(spatial_dims=3, kernel_size=3, stride=1, norm_name=norm_name, res_block=True)
        self.encoder1 = UnetrBasicBlock(in_channels=in_channels, out_channels=f, **kw)
        self.encoder2 = UnetrBasicBlock(in_channels=f, out_channels=f, **kw)
        self.encoder3 = UnetrBasicBlock(in_channels=2 * f, out_channels=2 * f, **kw)
        self.encoder4 = UnetrBasicBlock(in_channels=4 * f, out_channels=4 * f, **kw)
        self.encoder10 = UnetrBasicBlock(in_channels=16 * f, out_channels=16 * f, **kw)
        up = dict(spatial_dims=3, kernel_size=3, upsample_kernel_size=2, norm_name=norm_name, res_block=True)
        self.decoder5 = UnetrUpBlock(in_channels=16 * f, out_channels=8 * f, **up)
        self.decoder4 = UnetrUpBlock(in_channels=8 * f, out_channels=4 * f, **up)
        self.decoder3 = UnetrUpBlock(in_channels=4 * f, out_channels=2 * f, **up)
        self.decoder2 = UnetrUpBlock(in_channels=2 * f, out_channels=f, **up)
        self.decoder1 = UnetrUpBlock(in_channels=f, out_channels=f, **up)
        self.out = UnetOutBlock(spatial_dims=3, in_channels=f, out_channels=out_channels)

    def load_from(self, weights):
        """Copy a self-supervised Swin-ViT checkpoint (`weights["state_dict"]`, keys `module.<swinViT path>`) into the trunk —
        /root/reference/model/dim3/swin_unetr.py:230-277: the patch embedding, every block of layers1..4 (SwinTransformerBlock
        .load_from, :610-643) and each stage's PatchMerging reduction + norm.  Missing keys raise KeyError, mismatched shapes
        RuntimeError, as the reference's copy_ calls do."""
        sd = weights["state_dict"]
        with torch.no_grad():
            self.swinViT.patch_embed.proj.weight.copy_(sd["module.patch_embed.proj.weight"])
            self.swinViT.patch_embed.proj.bias.copy_(sd["module.patch_embed.proj.bias"])
            for layer in ("layers1", "layers2", "layers3", "layers4"):
                stage = getattr(self.swinViT, layer)[0]
                for bname, block in stage.blocks.named_children():
                    block.load_from(weights, n_block=bname, layer=layer)
                for name in ("reduction.weight", "norm.weight", "norm.bias"):
                    mod, attr = name.split(".")
                    getattr(getattr(stage.downsample, mod), attr).copy_(sd[f"module.{layer}.0.downsample.{name}"])

    @eager_only
    def forward(self, x_in):
        dtype = Fn.compute_dtype()
        with torch.autocast(device_type=x_in.device.type, enabled=False):
            x_in = x_in.float()
            hs = [h.to(dtype) for h in self.swinViT(x_in, self.normalize)]
            enc0 = self.encoder1(x_in, dtype)
            enc1 = self.encoder2(hs[0])
            enc2 = self.encoder3(hs[1])
            enc3 = self.encoder4(hs[2])
            dec4 = self.encoder10(hs[4])
            dec3 = self.decoder5(dec4, hs[3])
            dec2 = self.decoder4(dec3, enc3)
            dec1 = self.decoder3(dec2, enc2)
            dec0 = self.decoder2(dec1, enc1)
            out = self.decoder1(dec0, enc0)
            return Fn.HeadFn.apply(out, self.out.conv.conv.weight, self.out.conv.conv.bias)
