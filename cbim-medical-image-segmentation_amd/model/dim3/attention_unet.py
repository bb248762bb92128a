"""AttentionUNet behind the reference's constructor signature and parameter names
(/root/reference/model/dim3/attention_unet.py:8-47, attention_unet_utils.py:6-66).

Attention gate on the kernels already in the library: the two 1x1 projections are implicit-GEMM launches with
the InstanceNorm statistics (eps 1e-5, nn.InstanceNorm3d default) from their epilogues, relu(IN(g1)+IN(x1)) is the
post-norm residual-tail kernel, the 1-channel psi projection is the 1x1 head kernel, its InstanceNorm+sigmoid act on
one float per voxel (torch elementwise ops), and x*psi is ``functional.GateFn``."""
import torch
import torch.nn as nn

from ... import functional as Fn
from ...functional import eager_only
from ...ops import ACT
from .conv_layers import BasicBlock
from .unet_utils import down_block, inconv
from .utils import get_block, get_norm

_EPS = 1e-5


def _k3(k):
    return [k] * 3 if isinstance(k, int) else list(k)


class AttentionBlock(nn.Module):
    def __init__(self, g_ch, l_ch, int_ch):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv3d(g_ch, int_ch, kernel_size=1, bias=False), nn.Identity())
        self.W_x = nn.Sequential(nn.Conv3d(l_ch, int_ch, kernel_size=1, bias=False), nn.Identity())
        self.psi = nn.Sequential(nn.Conv3d(int_ch, 1, kernel_size=1, bias=False), nn.Identity(), nn.Identity())
        self.relu = nn.Identity()

    def forward(self, g, x):
        """g: upsampled low-res feature, x: encoder feature (channels-last tensors) -> x * psi."""
        g1, sg = Fn.NormConvFn.apply(g, None, self.W_g[0].weight, 0, None, True, None, _EPS)
        x1, sx = Fn.NormConvFn.apply(x, None, self.W_x[0].weight, 0, None, True, None, _EPS)
        p = Fn.ResNormFn.apply(g1, sg, x1, sx, ACT["relu"])                       # relu(IN(g1) + IN(x1))
        zero = torch.zeros(1, dtype=torch.float32, device=p.device)
        z = Fn.HeadFn.apply(p, self.psi[0].weight, zero)                          # [N,1,D,H,W] float32
        zf = z.flatten(2)
        var, mean = torch.var_mean(zf, dim=2, unbiased=False, keepdim=True)
        psi = torch.sigmoid((zf - mean) * torch.rsqrt(var + _EPS)).reshape(z.shape[0], *z.shape[2:])
        return Fn.GateFn.apply(x, psi)


class attention_up_block(nn.Module):
    def __init__(self, in_ch, out_ch, num_block, block=BasicBlock, kernel_size=(3, 3, 3), up_scale=(2, 2, 2), norm="in"):
        super().__init__()
        self.conv_ch = nn.Conv3d(in_ch, out_ch, kernel_size=1)      # declared but never used by the reference (:42)
        self.up_scale = _k3(up_scale)
        self.attn = AttentionBlock(in_ch, out_ch, out_ch // 2)
        k = _k3(kernel_size)
        mods = [block(in_ch + out_ch, out_ch, kernel_size=k, norm=norm)]
        for _ in range(num_block - 1):
            mods.append(block(out_ch, out_ch, kernel_size=k, norm=norm))
        self.conv = nn.Sequential(*mods)

    def forward(self, low: Fn.FMap, skip: Fn.FMap) -> Fn.FMap:
        empty = skip.t[..., :0]
        x1 = Fn.UpCatFn.apply(low.t, empty, True)                   # trilinear(align_corners) to the skip's size
        x2 = self.attn(x1, skip.t)
        f = Fn.FMap(torch.cat([x2, x1], dim=-1), None)
        for m in self.conv:
            f = m(f)
        return f


class AttentionUNet(nn.Module):
    def __init__(self, in_ch, base_ch, scale, kernel_size, num_classes=1, block="SingleConv", pool=True, norm="bn"):
        super().__init__()
        num_block = 2
        block = get_block(block)
        norm = get_norm(norm)      # in | bn | ln reach the BLOCKS; the gates keep nn.InstanceNorm3d (attention_unet_utils.py:10-21)
        b = base_ch
        self.inc = inconv(in_ch, b, block=block, kernel_size=kernel_size[0], norm=norm)
        self.down1 = down_block(b, 2 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[0], kernel_size=kernel_size[1], norm=norm)
        self.down2 = down_block(2 * b, 4 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[1], kernel_size=kernel_size[2], norm=norm)
        self.down3 = down_block(4 * b, 8 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[2], kernel_size=kernel_size[3], norm=norm)
        self.down4 = down_block(8 * b, 10 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[3], kernel_size=kernel_size[4], norm=norm)
        self.up1 = attention_up_block(10 * b, 8 * b, num_block=num_block, block=block, up_scale=scale[3], kernel_size=kernel_size[3], norm=norm)
        self.up2 = attention_up_block(8 * b, 4 * b, num_block=num_block, block=block, up_scale=scale[2], kernel_size=kernel_size[2], norm=norm)
        self.up3 = attention_up_block(4 * b, 2 * b, num_block=num_block, block=block, up_scale=scale[1], kernel_size=kernel_size[1], norm=norm)
        self.up4 = attention_up_block(2 * b, b, num_block=num_block, block=block, up_scale=scale[0], kernel_size=kernel_size[0], norm=norm)
        self.outc = nn.Conv3d(b, num_classes, kernel_size=1)

    @eager_only
    def forward(self, x):
        dtype = Fn.compute_dtype()
        with torch.autocast(device_type=x.device.type, enabled=False):
            x1 = self.inc(x, dtype)
            x2 = self.down1(x1)
            x3 = self.down2(x2)
            x4 = self.down3(x3)
            x5 = self.down4(x4)
            out = self.up1(x5, x4)
            out = self.up2(out, x3)
            out = self.up3(out, x2)
            out = self.up4(out, x1)
            return Fn.HeadFn.apply(out.t, self.outc.weight, self.outc.bias)
