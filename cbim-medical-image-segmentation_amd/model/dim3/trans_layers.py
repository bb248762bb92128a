"""The 192-token fusion transformer of SemanticMapFusion with the reference's parameter names
(/root/reference/model/dim3/trans_layers.py:16-118).  It sees 3 x 64 map tokens of width 320
(0.5 GFLOP) — far below one kernel launch worth of work per op — so it is expressed
with torch's own Linear/LayerNorm/softmax ops (hipBLASLt on the GPU) rather than hand-written kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Mlp(nn.Module):
    def __init__(self, in_dim, hid_dim=None, out_dim=None, act=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, hid_dim or in_dim)
        self.act = act()
        self.fc2 = nn.Linear(hid_dim or in_dim, out_dim or in_dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


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
        q, k, v = (t.reshape(B, Lt, self.heads, -1).transpose(1, 2) for t in self.to_qkv(x).chunk(3, dim=-1))
        p = F.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        return self.to_out(torch.matmul(p, v).transpose(1, 2).reshape(B, Lt, -1))


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
