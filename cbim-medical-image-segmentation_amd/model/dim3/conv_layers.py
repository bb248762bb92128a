"""Blocks of the 3D UNet family with the reference's module/parameter names
(/root/reference/model/dim3/conv_layers.py) executed by the fused HIP blocks in
``cbim_amd.functional``.

``nn.Conv3d`` instances are kept purely as PARAMETER HOLDERS (same ``weight``/``bias`` keys,
shapes and default initialisation as the reference, so ``state_dict``s interchange and the
same torch seed draws the same weights); their ``forward`` is never called.
InstanceNorm3d(eps=1e-4, affine=False) and the activation are parameter-free and live inside
the kernels (conv_layers.py:40-43).
"""
import torch.nn as nn

from ... import functional as Fn
from ...ops import ACT


def _k3(k):
    return [k] * 3 if isinstance(k, int) else list(k)


class ConvNormAct(nn.Module):
    """Parameter holder for one conv + (norm, act) description (conv_layers.py:16-53)."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, norm="in", act="relu", preact=False):
        super().__init__()
        if stride not in (1, [1, 1, 1], (1, 1, 1)):
            raise NotImplementedError("cbim_amd: strided ConvNormAct (pool=False) is not built")
        k = _k3(kernel_size)
        self.conv = nn.Conv3d(in_ch, out_ch, kernel_size=k, stride=1, padding=[i // 2 for i in k], bias=False)
        self.norm = nn.Identity()   # InstanceNorm3d(eps=1e-4): no parameters, fused into the kernels
        self.act = nn.Identity()
        self.act_code = ACT[act]
        self.preact = preact


class SingleConv(nn.Module):
    """act(IN(conv(x))) — conv_layers.py:56-68."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=1, norm="in", act="relu", preact=False):
        super().__init__()
        if preact:
            raise NotImplementedError("cbim_amd: SingleConv(preact=True) is not built")
        self.conv = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=False)

    def forward(self, f: Fn.FMap, need_dx=True) -> Fn.FMap:
        y = Fn.SingleConvFn.apply(f.t, self.conv.conv.weight, self.conv.act_code, need_dx)
        return Fn.FMap(y, None)


class BasicBlock(nn.Module):
    """conv2(conv1(x)) + shortcut(x), every ConvNormAct pre-activated — conv_layers.py:71-94."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=1, norm="in", act="relu", preact=True):
        super().__init__()
        if not preact:
            raise NotImplementedError("cbim_amd: BasicBlock(preact=False) is not built")
        self.conv1 = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=True)
        self.conv2 = ConvNormAct(out_ch, out_ch, kernel_size, stride=1, norm=norm, act=act, preact=True)
        self.shortcut = nn.Sequential()
        if in_ch != out_ch:   # a FULL k-sized pre-act ConvNormAct, not 1x1 (conv_layers.py:83-84)
            self.shortcut = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=True)

    def forward(self, f: Fn.FMap, want_out_stats=True) -> Fn.FMap:
        f = Fn.ensure_stats(f)
        wsc = self.shortcut.conv.weight if isinstance(self.shortcut, ConvNormAct) else None
        out, so = Fn.BasicBlockFn.apply(f.t, f.stats, self.conv1.conv.weight, self.conv2.conv.weight, wsc,
                                        self.conv1.act_code, want_out_stats)
        return Fn.FMap(out, so if want_out_stats else None)
