"""inconv / down_block / up_block with the reference's names
(/root/reference/model/dim3/unet_utils.py:7-75)."""
import torch.nn as nn

from ... import functional as Fn
from .conv_layers import BasicBlock, SingleConv


def _k3(k):
    return [k] * 3 if isinstance(k, int) else list(k)


class _PoolSlot(nn.Module):
    """Occupies index 0 of down_block.conv like nn.MaxPool3d does (no parameters)."""

    def __init__(self, scale):
        super().__init__()
        self.scale = tuple(_k3(scale))

    def forward(self, f: Fn.FMap) -> Fn.FMap:
        return Fn.FMap(Fn.MaxPoolFn.apply(f.t, self.scale), None)


class inconv(nn.Module):
    """raw Conv3d(in, out, k, pad k//2, bias=False) then one block (unet_utils.py:14-21)."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), block=BasicBlock, norm="in"):
        super().__init__()
        k = _k3(kernel_size)
        self.conv1 = nn.Conv3d(in_ch, out_ch, kernel_size=k, padding=[i // 2 for i in k], bias=False)
        self.conv2 = block(out_ch, out_ch, kernel_size=k, norm=norm)

    def forward(self, x, dtype) -> Fn.FMap:
        y = Fn.StemFn.apply(x, self.conv1.weight, dtype)
        return self.conv2(Fn.FMap(y, None))


class down_block(nn.Module):
    """MaxPool3d(down_scale) -> block(in,out) -> block(out,out) (unet_utils.py:35-46)."""

    def __init__(self, in_ch, out_ch, num_block, block=BasicBlock, kernel_size=(3, 3, 3), down_scale=(2, 2, 2),
                 pool=True, norm="in"):
        super().__init__()
        k = _k3(kernel_size)
        if pool:
            mods = [_PoolSlot(down_scale), block(in_ch, out_ch, kernel_size=k, norm=norm)]
        else:   # unet_utils.py:38-39 of the reference: the first block strides instead of a MaxPool3d in front of it
            mods = [block(in_ch, out_ch, stride=_k3(down_scale), kernel_size=k, norm=norm)]
        for _ in range(num_block - 1):
            mods.append(block(out_ch, out_ch, kernel_size=k, norm=norm))
        self.conv = nn.Sequential(*mods)

    def forward(self, f: Fn.FMap) -> Fn.FMap:
        for m in self.conv:
            f = m(f)
        return f


class up_block(nn.Module):
    """trilinear(align_corners) to the skip size -> cat([skip, up]) -> block -> block
    (unet_utils.py:62-75)."""

    def __init__(self, in_ch, out_ch, num_block, block=BasicBlock, kernel_size=(3, 3, 3), up_scale=(2, 2, 2),
                 norm="in"):
        super().__init__()
        k = _k3(kernel_size)
        self.up_scale = _k3(up_scale)
        mods = [block(in_ch + out_ch, out_ch, kernel_size=k, norm=norm)]
        for _ in range(num_block - 1):
            mods.append(block(out_ch, out_ch, kernel_size=k, norm=norm))
        self.conv = nn.Sequential(*mods)

    def forward(self, low: Fn.FMap, skip: Fn.FMap) -> Fn.FMap:
        first = self.conv[0]
        if Fn.fused_up_block(first):
            # a = relu(IN([skip | up(low)])) straight from `low` and `skip`: the concatenation is never stored
            skip = Fn.ensure_stats(skip)
            want = len(self.conv) > 1
            out, so = Fn.UpBlockFirstFn.apply(low.t, skip.t, skip.stats, first.conv1.conv.weight, first.conv2.conv.weight,
                                              first.shortcut.conv.weight, first.conv1.act_code, want, True)
            f = Fn.FMap(out, so if want else None)
            rest = list(self.conv)[1:]
        else:
            f = Fn.FMap(*Fn.UpCatFn.apply(low.t, skip.t, True, True))   # concat + its InstanceNorm statistics in one pass
            rest = list(self.conv)
        for m in rest:
            f = m(f)
        return f
