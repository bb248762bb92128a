"""UNet++ behind the reference's constructor signature and parameter names
(/root/reference/model/dim3/unetpp.py:8-90): the dense nested skip pathways are compositions of the blocks,
pooling and trilinear-upsample+concat kernels of the UNet family."""
import torch
import torch.nn as nn

from ... import functional as Fn
from ...functional import eager_only
from .utils import get_block, get_norm


def _k3(k):
    return [k] * 3 if isinstance(k, int) else list(k)


class UNetPlusPlus(nn.Module):
    def __init__(self, in_ch, base_ch, scale, kernel_size, num_classes=1, block="SingleConv", norm="bn"):
        super().__init__()
        num_block = 2
        block = get_block(block)
        norm = get_norm(norm)      # in | bn | ln: the blocks' composed paths (round 5)
        n_ch = [base_ch, base_ch * 2, base_ch * 4, base_ch * 8, base_ch * 10]
        self.scale = [tuple(_k3(s)) for s in scale]
        for i in range(4):   # parameter-free slots with the reference's attribute names
            setattr(self, f"pool{i}", nn.Identity())
            setattr(self, f"up{i}", nn.Identity())
        mk = lambda cin, cout, k: self.make_layer(cin, cout, num_block, block, kernel_size=k, norm=norm)
        ks = kernel_size
        self.conv0_0 = mk(in_ch, n_ch[0], ks[0])
        self.conv1_0 = mk(n_ch[0], n_ch[1], ks[1])
        self.conv2_0 = mk(n_ch[1], n_ch[2], ks[2])
        self.conv3_0 = mk(n_ch[2], n_ch[3], ks[3])
        self.conv4_0 = mk(n_ch[3], n_ch[4], ks[4])
        self.conv0_1 = mk(n_ch[0] + n_ch[1], n_ch[0], ks[0])
        self.conv1_1 = mk(n_ch[1] + n_ch[2], n_ch[1], ks[1])
        self.conv2_1 = mk(n_ch[2] + n_ch[3], n_ch[2], ks[2])
        self.conv3_1 = mk(n_ch[3] + n_ch[4], n_ch[3], ks[3])
        self.conv0_2 = mk(n_ch[0] * 2 + n_ch[1], n_ch[0], ks[0])
        self.conv1_2 = mk(n_ch[1] * 2 + n_ch[2], n_ch[1], ks[1])
        self.conv2_2 = mk(n_ch[2] * 2 + n_ch[3], n_ch[2], ks[2])
        self.conv0_3 = mk(n_ch[0] * 3 + n_ch[1], n_ch[0], ks[0])
        self.conv1_3 = mk(n_ch[1] * 3 + n_ch[2], n_ch[1], ks[1])
        self.conv0_4 = mk(n_ch[0] * 4 + n_ch[1], n_ch[0], ks[0])
        self.output = nn.Conv3d(n_ch[0], num_classes, kernel_size=1)

    def make_layer(self, in_ch, out_ch, num_block, block, kernel_size, norm):
        blocks = [block(in_ch, out_ch, kernel_size=kernel_size, norm=norm)]
        for _ in range(num_block - 1):
            blocks.append(block(out_ch, out_ch, kernel_size=kernel_size, norm=norm))
        return nn.Sequential(*blocks)

    @staticmethod
    def _run(layer, f):
        for m in layer:
            f = m(f)
        return f

    def _node(self, layer, skips, low):
        """conv(cat([*skips, up(low)], 1)) — unetpp.py:54-73 (nn.Upsample(scale_factor, trilinear, align_corners=True)
        lands exactly on the skip's size for the even extents the reference supports)."""
        sk = skips[0].t if len(skips) == 1 else torch.cat([s.t for s in skips], dim=-1)
        return self._run(layer, Fn.FMap(*Fn.UpCatFn.apply(low.t, sk, True, True)))   # concat + statistics in one pass

    @eager_only
    def forward(self, x):
        dtype = Fn.compute_dtype()
        with torch.autocast(device_type=x.device.type, enabled=False):
            pool = lambda f, i: Fn.FMap(Fn.MaxPoolFn.apply(f.t, self.scale[i]), None)
            x0_0 = self._run(self.conv0_0[1:], self.conv0_0[0].forward_input(x, dtype))
            x1_0 = self._run(self.conv1_0, pool(x0_0, 0))
            x0_1 = self._node(self.conv0_1, [x0_0], x1_0)
            x2_0 = self._run(self.conv2_0, pool(x1_0, 1))
            x1_1 = self._node(self.conv1_1, [x1_0], x2_0)
            x0_2 = self._node(self.conv0_2, [x0_0, x0_1], x1_1)
            x3_0 = self._run(self.conv3_0, pool(x2_0, 2))
            x2_1 = self._node(self.conv2_1, [x2_0], x3_0)
            x1_2 = self._node(self.conv1_2, [x1_0, x1_1], x2_1)
            x0_3 = self._node(self.conv0_3, [x0_0, x0_1, x0_2], x1_2)
            x4_0 = self._run(self.conv4_0, pool(x3_0, 3))
            x3_1 = self._node(self.conv3_1, [x3_0], x4_0)
            x2_2 = self._node(self.conv2_2, [x2_0, x2_1], x3_1)
            x1_3 = self._node(self.conv1_3, [x1_0, x1_1, x1_2], x2_2)
            x0_4 = self._node(self.conv0_4, [x0_0, x0_1, x0_2, x0_3], x1_3)
            return Fn.HeadFn.apply(x0_4.t, self.output.weight, self.output.bias)
