"""3D UNet / ResUNet behind the reference's constructor signature and parameter names
(/root/reference/model/dim3/unet.py:12-64); forward/backward run on hand-written gfx950
kernels (``cbim_amd.functional``).  forward(x[B,C,D,H,W] fp32 NCDHW) -> logits[B,classes,D,H,W] fp32.
"""
import torch
import torch.nn as nn

from ... import functional as Fn
from ...functional import eager_only
from .unet_utils import down_block, inconv, up_block
from .utils import get_block, get_norm


class UNet(nn.Module):
    def __init__(self, in_ch, base_ch, scale=[2, 2, 2, 2], kernel_size=[3, 3, 3, 3], num_classes=1,
                 block="ConvNormAct", pool=True, norm="bn"):
        super().__init__()
        num_block = 2
        block = get_block(block)   # KeyError on the reference's (unusable) default, like the reference
        norm = get_norm(norm)
        b = base_ch
        self.inc = inconv(in_ch, b, block=block, kernel_size=kernel_size[0], norm=norm)
        self.down1 = down_block(b, 2 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[0],
                                kernel_size=kernel_size[1], norm=norm)
        self.down2 = down_block(2 * b, 4 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[1],
                                kernel_size=kernel_size[2], norm=norm)
        self.down3 = down_block(4 * b, 8 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[2],
                                kernel_size=kernel_size[3], norm=norm)
        self.down4 = down_block(8 * b, 10 * b, num_block=num_block, block=block, pool=pool, down_scale=scale[3],
                                kernel_size=kernel_size[4], norm=norm)
        self.up1 = up_block(10 * b, 8 * b, num_block=num_block, block=block, up_scale=scale[3],
                            kernel_size=kernel_size[3], norm=norm)
        self.up2 = up_block(8 * b, 4 * b, num_block=num_block, block=block, up_scale=scale[2],
                            kernel_size=kernel_size[2], norm=norm)
        self.up3 = up_block(4 * b, 2 * b, num_block=num_block, block=block, up_scale=scale[1],
                            kernel_size=kernel_size[1], norm=norm)
        self.up4 = up_block(2 * b, b, num_block=num_block, block=block, up_scale=scale[0],
                            kernel_size=kernel_size[0], norm=norm)
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
