"""The 192-token fusion transformer of SemanticMapFusion with the reference's parameter names
(/root/reference/model/dim3/trans_layers.py:16-118).  It sees 3 x 64 map tokens of width 320
(0.5 GFLOP) — far below one kernel launch worth of work per op — so it is expressed
with torch's own LayerNorm / softmax / matmul ops rather than hand-written kernels; its Linears run on the
engine's row GEMM in its fp32-exact form (bf16 hi + lo fragments of rows and weights) in the bf16 engine mode —
round 6: hipBLASLt picked 256-row macro tiles for the 216 tokens of config/acdc (60 - 80 us per GEMM).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _lin(lin: nn.Linear, x):
    """nn.Linear over the float32 token rows: the engine's fp32-exact row GEMM in the bf16 engine mode, torch's own otherwise"""
    from ... import _lib
    from ... import functional as Fn
    w = lin.weight
    if (Fn.compute_dtype() == torch.bfloat16 and x.dtype == torch.float32 and w.dtype == torch.float32
            and x.device.type == ("cpu" if _lib.backend() == "emu" else "cuda") and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
            and w.shape[1] <= 4096):
        return Fn.token_linear(x, w, lin.bias, out_dtype=torch.float32, exact=True)
    return lin(x)


class Mlp(nn.Module):
    def __init__(self, in_dim, hid_dim=None, out_dim=None, act=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, hid_dim or in_dim)
        self.act = act()
        self.fc2 = nn.Linear(hid_dim or in_dim, out_dim or in_dim)

    def forward(self, x):
        return _lin(self.fc2, self.act(_lin(self.fc1, x)))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class Attention(nn.Module):
    """softmax(q k^T / sqrt(d)) v with the '(heads dim_head)' channel split (trans_layers.py:60-68)."""

    def __init__(self, dim, heads, dim_head, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("cbim_amd: dropout in the map-fusion transformer is not built (0 in every config)")
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, dim_head * heads * 3, bias=False)
        self.to_out = nn.Linear(dim_head * heads, dim)

    def forward(self, x):
        B, Lt, _ = x.shape
        q, k, v = (t.reshape(B, Lt, self.heads, -1).transpose(1, 2) for t in _lin(self.to_qkv, x).chunk(3, dim=-1))
        p = F.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        return _lin(self.to_out, torch.matmul(p, v).transpose(1, 2).reshape(B, Lt, -1))


class TransformerBlock(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([PreNorm(dim, Attention(dim, heads, dim_head, attn_drop, proj_drop)),
                           PreNorm(dim, Mlp(dim, mlp_dim, dim))])
            for _ in range(depth)])

    def forward(self, x):
        for attn, ffn in self.layers:
            x = attn(x) + x
            x = ffn(x) + x
        return x


class LayerNorm(nn.Module):
    """The reference's channels-first LayerNorm (trans_layers.py:120-149), what `norm: ln` puts into ConvNormAct
    (model/dim3/utils.py:15-21, built with eps 1e-4 by conv_layers.py:40-42): per voxel, over the channel axis, with a per-channel
    affine.  A parameter holder with the reference's state_dict keys — the engine's tensors are channels-last, so the arithmetic is
    the token-row LayerNorm kernel (functional.LayerNormFn); forward() takes an NCDHW tensor like the reference's."""

    def __init__(self, normalized_shape, eps=1e-5, data_format="channels_first"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format = eps, data_format
        self.normalized_shape = (normalized_shape,)

    def rows(self, t):
        """channels-last tensor [..., C] -> LN over C (the engine-side call)"""
        from ... import functional as Fn
        y = Fn.layer_norm(t if t.dtype == torch.float32 else t.float(), self.weight, self.bias, self.eps, out_dtype=t.dtype)
        return y

    def forward(self, x):
        if self.data_format == "channels_last":
            return self.rows(x)
        C = int(x.shape[1])
        if C < 4 or C % 4:
            # a raw network input (1 - 3 channels: UNet++ conv0_0 with a pre-activated first block): the reference's own formula
            # on the device, a few elementwise torch ops over one volume (the row kernel takes channel counts in multiples of 4)
            u = x.mean(1, keepdim=True)
            v = (x - u).pow(2).mean(1, keepdim=True)
            shape = (1, -1) + (1,) * (x.dim() - 2)
            return self.weight.view(shape) * ((x - u) / torch.sqrt(v + self.eps)) + self.bias.view(shape)
        perm = [0] + list(range(2, x.dim())) + [1]
        inv = [0, x.dim() - 1] + list(range(1, x.dim() - 1))
        return self.rows(x.permute(perm).contiguous()).permute(inv)
