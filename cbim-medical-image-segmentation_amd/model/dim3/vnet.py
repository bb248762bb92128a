"""VNet behind the reference's constructor signature and parameter names (/root/reference/model/dim3/vnet.py:22-182,
dispatched at /root/reference/model/utils.py:70-74).  forward(x[B,C,D,H,W] fp32 NCDHW) -> logits[B,classes,D,H,W] fp32.

What runs where: the 5x5x5 convolutions as sums of (1,5,5) implicit-GEMM launches over D-shifted slices (`ConvSlicesFn`), the
strided / transposed k = s convolutions as space-to-depth / depth-to-space + a 1x1x1 implicit GEMM, ContBatchNorm3d (batch
statistics ALWAYS, vnet.py:22-33) + ELU on the InstanceNorm streaming kernels with the moments pooled over the batch
(`BatchNormActFn`), the ELU after the residual adds as one streaming pass (`ActFn`).  Residual adds, bias adds, the channel
concatenation and the Dropout3d channel mask (drawn with torch's RNG exactly as `F.dropout3d` draws it: one Bernoulli value per
(sample, channel), vnet.py:83-84,105-110 — active in training mode only) are elementwise ATen ops.  The 4-class tail of the
output transition (BatchNorm + ELU + 1x1x1 conv on `classes` channels, vnet.py:128-138) runs on the same kernels with the
channels zero-padded to one 16-byte chunk; the 1x1x1 convolution is the head kernel, which writes the NCDHW float32 logits.  ELU only (`elu=True`, what `get_model` builds);
PReLU raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as Fn
from ...functional import eager_only
from ...ops import ACT

_ELU = ACT["elu"]


def dropout3d_mask(n, c, p, training, device):
    """The multiplier F.dropout3d applies per (sample, channel): Bernoulli(1 - p) / (1 - p), drawn like ATen's feature dropout
    does (a [N, C, 1, 1, 1] noise tensor).  Tests replace this function to inject the masks of a golden run."""
    if not training or p == 0.0:
        return None
    return F.dropout3d(torch.ones(n, c, 1, 1, 1, device=device), p, True).view(n, c)


def _drop(x, p, training):
    m = dropout3d_mask(int(x.shape[0]), int(x.shape[-1]), p, training, x.device)
    return x if m is None else x * m.view(m.shape[0], 1, 1, 1, m.shape[1]).to(x.dtype)


def _k3(s):
    return [s] * 3 if isinstance(s, int) else list(s)


class ContBatchNorm3d(nn.modules.batchnorm._BatchNorm):
    """vnet.py:22-33: batch statistics in every mode (F.batch_norm(..., training=True))."""

    def _check_input_dim(self, input):
        if input.dim() != 5:
            raise ValueError("expected 5D input (got {}D input)".format(input.dim()))

    def forward(self, t, act=0):
        """t: channels-last [N,D,H,W,C] engine tensor -> act(batch_norm(t))."""
        return Fn.BatchNormActFn.apply(t, self.weight, self.bias, self.running_mean, self.running_var,
                                       0.1 if self.momentum is None else self.momentum, self.eps, act)


def ELUCons(elu, nchan):
    if not elu:
        raise NotImplementedError("cbim_amd: VNet with PReLU (elu=False) is not built; get_model() constructs elu=True")
    return nn.ELU(inplace=True)          # parameter-free: occupies the reference's attribute slot


class LUConv(nn.Module):
    def __init__(self, nchan, elu):
        super().__init__()
        self.relu1 = ELUCons(elu, nchan)
        self.conv1 = nn.Conv3d(nchan, nchan, kernel_size=5, padding=2)
        self.bn1 = ContBatchNorm3d(nchan)

    def forward(self, t):
        return self.bn1(Fn.ConvSlicesFn.apply(t, self.conv1.weight, self.conv1.bias), _ELU)


def _make_nConv(nchan, depth, elu):
    return nn.Sequential(*[LUConv(nchan, elu) for _ in range(depth)])


class InputTransition(nn.Module):
    def __init__(self, inChans, outChans, elu):
        super().__init__()
        self.conv1 = nn.Conv3d(inChans, outChans, kernel_size=5, padding=2)
        self.bn1 = ContBatchNorm3d(outChans)
        self.relu1 = ELUCons(elu, outChans)
        self.inChans, self.outChans = inChans, outChans

    def forward(self, x, dtype):
        t = Fn.StemFn.apply(x, self.conv1.weight, dtype)               # NCDHW fp32 -> channels-last
        t = Fn.BiasAddFn.apply(t, self.conv1.bias)
        out = self.bn1(t, 0)
        num = int(self.outChans / self.inChans)
        x16 = x.permute(0, 2, 3, 4, 1).repeat(1, 1, 1, 1, num).to(out.dtype)   # x.repeat(1, num, 1, 1, 1), channels-last
        return Fn.ActFn.apply(out + x16, _ELU)


class DownTransition(nn.Module):
    def __init__(self, inChans, nConvs, elu, scale=2, dropout=False):
        super().__init__()
        outChans = 2 * inChans
        self.scale = _k3(scale)
        self.down_conv = nn.Conv3d(inChans, outChans, kernel_size=scale, stride=scale)
        self.bn1 = ContBatchNorm3d(outChans)
        self.do1 = nn.Dropout3d() if dropout else None                 # (reference: `passthrough` / nn.Dropout3d, no parameters)
        self.relu1 = ELUCons(elu, outChans)
        self.relu2 = ELUCons(elu, outChans)
        self.ops = _make_nConv(outChans, nConvs, elu)

    def forward(self, t):
        w = self.down_conv.weight                                      # [Cout, Cin, sD, sH, sW], stride = kernel
        w_eq = w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1, 1, 1, 1)   # channel order of k_space_to_depth: ((i sH + j) sW + k) C + c
        m = Fn.SpaceToDepthFn.apply(t, tuple(self.scale))
        d, _ = Fn.NormConvFn.apply(m, None, w_eq, 0, None, False, None, 1e-5)
        d = Fn.BiasAddFn.apply(d, self.down_conv.bias)
        down = self.bn1(d, _ELU)
        out = _drop(down, self.do1.p, self.training) if self.do1 is not None else down
        for m_ in self.ops:
            out = m_(out)
        return Fn.ActFn.apply(out + down, _ELU)


class UpTransition(nn.Module):
    def __init__(self, inChans, outChans, nConvs, elu, scale=2, dropout=False):
        super().__init__()
        self.scale = _k3(scale)
        self.up_conv = nn.ConvTranspose3d(inChans, outChans // 2, kernel_size=scale, stride=scale)
        self.bn1 = ContBatchNorm3d(outChans // 2)
        self.do1 = nn.Dropout3d() if dropout else None
        self.do2 = nn.Dropout3d()
        self.relu1 = ELUCons(elu, outChans // 2)
        self.relu2 = ELUCons(elu, outChans)
        self.ops = _make_nConv(outChans, nConvs, elu)

    def forward(self, t, skip):
        out = _drop(t, self.do1.p, self.training) if self.do1 is not None else t
        skipdo = _drop(skip, self.do2.p, self.training)
        w = self.up_conv.weight                                        # [Cin, Cout, sD, sH, sW]
        cout = int(w.shape[1])
        w_eq = w.permute(2, 3, 4, 1, 0).reshape(-1, w.shape[0], 1, 1, 1)   # rows ((i sH + j) sW + k) Cout + co
        u, _ = Fn.NormConvFn.apply(out, None, w_eq, 0, None, False, None, 1e-5)
        u = Fn.DepthToSpaceFn.apply(u, tuple(self.scale))
        u = Fn.BiasAddFn.apply(u, self.up_conv.bias)
        assert int(u.shape[-1]) == cout
        u = self.bn1(u, _ELU)
        xcat = torch.cat((u, skipdo), -1)
        out = xcat
        for m_ in self.ops:
            out = m_(out)
        return Fn.ActFn.apply(out + xcat, _ELU)


class OutputTransition(nn.Module):
    def __init__(self, inChans, outChans, elu, nll):
        super().__init__()
        self.conv1 = nn.Conv3d(inChans, outChans, kernel_size=5, padding=2)
        self.bn1 = ContBatchNorm3d(outChans)
        self.conv2 = nn.Conv3d(outChans, outChans, kernel_size=1)
        self.relu1 = ELUCons(elu, outChans)

    def forward(self, t):
        # the 5x5x5 convolution down to `classes` channels runs on the kernels with the output channels padded to a whole
        # 16-byte chunk (zero weight rows / bias: the padded channels are exactly 0 through BatchNorm [gamma 1, beta 0] and
        # ELU); BatchNorm + ELU on the norm kernels, the 1x1x1 convolution on the head kernel (zero weight columns for the
        # padding), which writes the NCDHW float32 logits.  (torch's conv3d on these 4-channel NCDHW planes would run MIOpen's
        # fallback weight-gradient kernel: 176 ms per step at the ACDC crop.)
        w, b = self.conv1.weight, self.conv1.bias
        cout = int(w.shape[0])
        cpc = 8 if t.dtype == torch.bfloat16 else 4
        pad = (-cout) % cpc
        bn = self.bn1
        gamma, beta, rm, rv = bn.weight, bn.bias, bn.running_mean, bn.running_var
        w2 = self.conv2.weight
        if pad:
            w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)
            b = torch.cat([b, b.new_zeros(pad)], 0)
            gamma = torch.cat([gamma, gamma.new_ones(pad)], 0)
            beta = torch.cat([beta, beta.new_zeros(pad)], 0)
            rm = torch.cat([rm, rm.new_zeros(pad)], 0)
            rv = torch.cat([rv, rv.new_ones(pad)], 0)
            w2 = torch.cat([w2, w2.new_zeros((int(w2.shape[0]), pad) + tuple(w2.shape[2:]))], 1)
        y = Fn.ConvSlicesFn.apply(t, w, b)
        y = Fn.BatchNormActFn.apply(y, gamma, beta, rm, rv, 0.1 if bn.momentum is None else bn.momentum, bn.eps, _ELU)
        if pad:
            with torch.no_grad():
                bn.running_mean.copy_(rm[:cout])
                bn.running_var.copy_(rv[:cout])
        return Fn.HeadFn.apply(y, w2, self.conv2.bias)


class VNet(nn.Module):
    def __init__(self, inChans, outChans, scale, baseChans=16, elu=True, nll=False):
        super().__init__()
        self.in_tr = InputTransition(inChans, baseChans, elu)
        self.down_tr32 = DownTransition(baseChans, 1, elu, scale=scale[0])
        self.down_tr64 = DownTransition(baseChans * 2, 2, elu, scale=scale[1])
        self.down_tr128 = DownTransition(baseChans * 4, 3, elu, dropout=True, scale=scale[2])
        self.down_tr256 = DownTransition(baseChans * 8, 2, elu, dropout=True, scale=scale[3])
        self.up_tr256 = UpTransition(baseChans * 16, baseChans * 16, 2, elu, dropout=True, scale=scale[3])
        self.up_tr128 = UpTransition(baseChans * 16, baseChans * 8, 2, elu, dropout=True, scale=scale[2])
        self.up_tr64 = UpTransition(baseChans * 8, baseChans * 4, 1, elu, scale=scale[1])
        self.up_tr32 = UpTransition(baseChans * 4, baseChans * 2, 1, elu, scale=scale[0])
        self.out_tr = OutputTransition(baseChans * 2, outChans, elu, nll)

    @eager_only
    def forward(self, x):
        dtype = Fn.compute_dtype()
        with torch.autocast(device_type=x.device.type, enabled=False):
            x = x.contiguous().float()
            out16 = self.in_tr(x, dtype)
            out32 = self.down_tr32(out16)
            out64 = self.down_tr64(out32)
            out128 = self.down_tr128(out64)
            out256 = self.down_tr256(out128)
            out = self.up_tr256(out256, out128)
            out = self.up_tr128(out, out64)
            out = self.up_tr64(out, out32)
            out = self.up_tr32(out, out16)
            return self.out_tr(out)
