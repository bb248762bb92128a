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
from ...ops import ACT, IN_EPS

_SE_KERNELS = True     # SEBlock excitation on the engine's own kernels (round 6); tests flip it for the two-sided check


def _k3(k):
    return [k] * 3 if isinstance(k, int) else list(k)


def _torch_act(code, z):
    """the activation of a ConvNormAct as a torch op (raw-input paths: a 1 - 3 channel volume)"""
    import torch.nn.functional as F
    if code == ACT["relu"]:
        return F.relu(z)
    if code == ACT["lrelu"]:
        return F.leaky_relu(z)          # nn.LeakyReLU() default slope 0.01 (model/dim3/utils.py:24-30)
    if code == ACT["gelu"]:
        return F.gelu(z)
    if code == ACT["swish"]:
        return F.silu(z)
    if code == 0:
        return z
    raise NotImplementedError(f"cbim_amd: activation code {code} on the raw input")


def _raw_input_conv(cna, a, dtype):
    """conv(a) of an NCDHW fp32 tensor with 1 - 7 channels whose producer HAS parameters (the pre-activation norm of `norm: bn | ln`
    on the network input): the stem kernel returns no input gradient, so the tensor goes channels-last, zero-padded to one 8-channel
    chunk, through the dense convolution (weight padded to match; autograd slices its gradient back)."""
    import torch.nn.functional as F
    c = int(a.shape[1])
    cp = (c + 7) // 8 * 8
    t = F.pad(a.permute(0, 2, 3, 4, 1), (0, cp - c)).to(dtype).contiguous()
    w = F.pad(cna.conv.weight, (0, 0, 0, 0, 0, 0, 0, cp - c))
    return Fn.NormConvFn.apply(t, None, w, 0, None, False, None, IN_EPS)[0]


def bn_act(bn, t, act_code=0):
    """act(BatchNorm3d(t)) on a channels-last tensor through the module `bn` (a parameter / buffer holder): batch statistics +
    running-statistics update in train(), the running statistics in eval() (or when they are not tracked: always batch
    statistics) — nn.BatchNorm3d.forward semantics"""
    batch = bn.training or bn.running_mean is None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    mom = bn.momentum
    if mom is None:                                   # cumulative moving average
        mom = 1.0 / float(bn.num_batches_tracked) if bn.num_batches_tracked is not None else 0.0
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return Fn.BatchNormActFn.apply(t, bn.weight, bn.bias, rm, rv, mom, bn.eps, act_code, batch)


def norm_rows(norm, t, act_code=0):
    """act(norm(t)) of a channels-last tensor through a `norm: bn | ln` module (nn.BatchNorm3d / trans_layers.LayerNorm)"""
    if isinstance(norm, nn.BatchNorm3d):
        return bn_act(norm, t, act_code)
    y = norm.rows(t)
    return y if act_code == 0 else Fn.ActFn.apply(y, act_code)


def make_norm(kind, ch):
    """`norm(ch)` as the reference's blocks build it OUTSIDE ConvNormAct (default eps 1e-5: medformer_utils.py:112-113,158):
    nn.Identity for `in` (no parameters; the arithmetic is fused into the kernels), else the parameter holder"""
    if kind == "bn":
        return nn.BatchNorm3d(ch)
    if kind == "ln":
        from .trans_layers import LayerNorm
        return LayerNorm(ch)
    return nn.Identity()


class ConvNormAct(nn.Module):
    """Parameter holder for one conv + (norm, act) description (conv_layers.py:16-53)."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, groups=1, norm="in", act="relu",
                 preact=False):
        super().__init__()
        self.stride = tuple(_k3(stride))
        if any(s not in (1, 2) for s in self.stride):
            raise NotImplementedError("cbim_amd: ConvNormAct strides other than 1 and 2 are not built")
        if self.stride != (1, 1, 1) and groups != 1:
            raise NotImplementedError("cbim_amd: strided grouped convolutions are not built")
        if groups not in (1, in_ch):
            raise NotImplementedError("cbim_amd: grouped convolutions other than depthwise are not built")
        k = _k3(kernel_size)
        if self.stride != (1, 1, 1):
            Fn._stride_plan(k, self.stride)      # unsupported (kernel, stride) pairs fail HERE, not at the first forward
        self.conv = nn.Conv3d(in_ch, out_ch, kernel_size=k, stride=self.stride, padding=[i // 2 for i in k], groups=groups,
                              bias=False)
        if norm == "bn":
            # `norm: bn` (model/dim3/utils.py:15-21 of the reference): nn.BatchNorm3d(eps=1e-4) over in_ch (pre-activation) or
            # out_ch (conv_layers.py:40-43) — a PARAMETER / BUFFER holder with the reference's state_dict keys; its arithmetic
            # runs on the affine norm kernels (functional.BatchNormActFn)
            self.norm = nn.BatchNorm3d(in_ch if preact else out_ch, eps=IN_EPS)
        elif norm == "ln":
            # `norm: ln`: the reference's channels-first LayerNorm(eps=1e-4) (trans_layers.py:120-149) — per voxel over the
            # channels, which in the engine's channels-last layout is the token-row LayerNorm kernel
            from .trans_layers import LayerNorm
            self.norm = LayerNorm(in_ch if preact else out_ch, eps=IN_EPS)
        elif norm in ("in", None, False, True):
            self.norm = nn.Identity()   # InstanceNorm3d(eps=1e-4): no parameters, fused into the kernels
        else:
            raise NotImplementedError(f"cbim_amd: norm '{norm}' is not built")
        self.norm_kind = norm if norm in ("bn", "ln") else "in"
        self.depthwise = groups != 1
        self.act = nn.Identity()
        self.act_code = ACT[act]
        self.preact = preact

    def _norm_act(self, t):
        """act(norm(t)) of the composed path: BatchNorm3d or the channel LayerNorm, one streaming pass each"""
        if self.norm_kind == "ln":
            y = self.norm.rows(t)
            return y if self.act_code == 0 else Fn.ActFn.apply(y, self.act_code)
        return self._bn_act(t)

    def _bn_act(self, t):
        return bn_act(self.norm, t, self.act_code)

    def apply_generic(self, t, res=None):
        """The composed (unfused) form of ConvNormAct.forward for `norm: bn`: conv(act(BN(t))) [+ res] (pre-activation) or
        act(BN(conv(t))) — the activated tensor is materialised by one streaming pass and the convolution reads it as it is."""
        if self.norm_kind == "in":
            raise RuntimeError("apply_generic is the BatchNorm / LayerNorm path")
        if self.preact:
            y = self.raw_conv(self._norm_act(t), res)
            return y
        y = self._norm_act(self.raw_conv(t))
        return y if res is None else y + res

    def raw_conv(self, a, res=None):
        """conv(a) [+ res] of a tensor used as it is: the stride-1 kernels directly, a stride of 2 through its space-to-depth form
        (functional.strided_conv; down_block(pool=False), unet_utils.py:36-39 of the reference)"""
        if self.depthwise:
            y = Fn.DWConvFn.apply(a, None, self.conv.weight, 0, False)[0]
            return y if res is None else y + res
        if self.stride == (1, 1, 1):
            return Fn.NormConvFn.apply(a, None, self.conv.weight, 0, res, False, None, IN_EPS)[0]
        y = Fn.strided_conv(a, self.conv.weight, self.stride)
        return y if res is None else y + res

    @property
    def strided(self):
        return self.stride != (1, 1, 1)


class SingleConv(nn.Module):
    """act(IN(conv(x))) — conv_layers.py:56-68."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=1, norm="in", act="relu", preact=False):
        super().__init__()
        if preact:
            raise NotImplementedError("cbim_amd: SingleConv(preact=True) is not built")
        self.conv = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=False)

    def forward(self, f: Fn.FMap, need_dx=True) -> Fn.FMap:
        if self.conv.norm_kind != "in":
            return Fn.FMap(self.conv.apply_generic(f.t), None)
        if self.conv.strided:                   # down_block(pool=False): act(IN(conv_stride(x)))
            z = self.conv.raw_conv(f.t)
            zs = Fn.ensure_stats(Fn.FMap(z, None)).stats
            return Fn.FMap(Fn.NormActFn.apply(z, zs, self.conv.act_code), None)
        y = Fn.SingleConvFn.apply(f.t, self.conv.conv.weight, self.conv.act_code, need_dx)
        return Fn.FMap(y, None)

    def forward_input(self, x, dtype) -> Fn.FMap:
        """First layer of a network: x is the NCDHW fp32 input (any channel count <= 16)."""
        z = Fn.StemFn.apply(x, self.conv.conv.weight, dtype)
        if self.conv.norm_kind != "in":
            return Fn.FMap(self.conv._norm_act(z), None)
        zs = Fn.ensure_stats(Fn.FMap(z, None)).stats
        return Fn.FMap(Fn.NormActFn.apply(z, zs, self.conv.act_code), None)


class BasicBlock(nn.Module):
    """conv2(conv1(x)) + shortcut(x), every ConvNormAct pre-activated — conv_layers.py:71-94."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=1, norm="in", act="relu", preact=True):
        super().__init__()
        if not preact:
            raise NotImplementedError("cbim_amd: BasicBlock(preact=False) is not built")
        self.conv1 = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=True)
        self.conv2 = ConvNormAct(out_ch, out_ch, kernel_size, stride=1, norm=norm, act=act, preact=True)
        self.shortcut = nn.Sequential()
        if in_ch != out_ch or self.conv1.strided:   # a FULL k-sized pre-act ConvNormAct, not 1x1 (conv_layers.py:83-84)
            self.shortcut = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=True)

    def cbim_grad_pairs(self):
        """(conv1, shortcut) weights: BasicBlockFn computes their gradients as ONE Cout-concatenated weight gradient —
        parallel.GradAllReduce lays their bucket slots out back to back so that the kernel writes both in place"""
        fused = isinstance(self.shortcut, ConvNormAct) and self.conv1.norm_kind == "in" and not self.conv1.strided
        return [(self.conv1.conv.weight, self.shortcut.conv.weight)] if fused else []

    def forward_input(self, x, dtype, want_out_stats=True) -> Fn.FMap:
        """First layer of a network (UNet++ conv0_0): x is the NCDHW fp32 input.  The pre-activation
        act(IN(x)) of the in_ch-channel input (2 M values at 128^3) is two torch elementwise ops; conv1 and the
        shortcut conv read it through the stem kernel."""
        import torch.nn.functional as F
        if self.conv1.norm_kind != "in":
            # `norm: bn | ln`: conv1 and the shortcut own SEPARATE norm parameters over the in_ch-channel input — the holders'
            # own forward (nn.BatchNorm3d: batch / running statistics and their update; the channels-first LayerNorm) on the raw
            # NCDHW volume, the activation as a torch op, then the stem kernel; conv2 + the residual add on the composed path
            if not isinstance(self.shortcut, ConvNormAct):
                raise NotImplementedError("cbim_amd: identity-shortcut BasicBlock on the raw network input is not built")
            xf = x.float()
            y1 = _raw_input_conv(self.conv1, _torch_act(self.conv1.act_code, self.conv1.norm(xf)), dtype)
            sc = _raw_input_conv(self.shortcut, _torch_act(self.shortcut.act_code, self.shortcut.norm(xf)), dtype)
            return Fn.FMap(self.conv2.apply_generic(y1, res=sc), None)
        if not isinstance(self.shortcut, ConvNormAct):
            raise NotImplementedError("cbim_amd: identity-shortcut BasicBlock on the raw network input is not built")
        if self.conv1.act_code != ACT["relu"]:
            raise NotImplementedError("cbim_amd: first-layer BasicBlock supports ReLU")
        a = F.relu(F.instance_norm(x.float(), eps=IN_EPS))
        y1 = Fn.StemFn.apply(a, self.conv1.conv.weight, dtype)
        s1 = Fn.ensure_stats(Fn.FMap(y1, None)).stats
        sc = Fn.StemFn.apply(a, self.shortcut.conv.weight, dtype)
        out, so = Fn.NormConvFn.apply(y1, s1, self.conv2.conv.weight, self.conv1.act_code, sc, want_out_stats, None, IN_EPS)
        return Fn.FMap(out, so if want_out_stats else None)

    def forward(self, f: Fn.FMap, want_out_stats=True) -> Fn.FMap:
        if self.conv1.norm_kind != "in":       # composed path: conv2(conv1(x)) + shortcut(x), the add in conv2's epilogue
            res = self.shortcut.apply_generic(f.t) if isinstance(self.shortcut, ConvNormAct) else f.t
            return Fn.FMap(self.conv2.apply_generic(self.conv1.apply_generic(f.t), res=res), None)
        f = Fn.ensure_stats(f)
        if self.conv1.strided:
            # down_block(pool=False): conv1 and the shortcut are strided and read the same act(IN(x)), written once
            act = self.conv1.act_code
            a = Fn.NormActFn.apply(f.t, f.stats, act)
            y1 = self.conv1.raw_conv(a)
            sc = self.shortcut.raw_conv(a)
            s1 = Fn.ensure_stats(Fn.FMap(y1, None)).stats
            out, so = Fn.NormConvFn.apply(y1, s1, self.conv2.conv.weight, act, sc, want_out_stats, None, IN_EPS)
            return Fn.FMap(out, so if want_out_stats else None)
        wsc = self.shortcut.conv.weight if isinstance(self.shortcut, ConvNormAct) else None
        out, so = Fn.BasicBlockFn.apply(f.t, f.stats, self.conv1.conv.weight, self.conv2.conv.weight, wsc,
                                        self.conv1.act_code, want_out_stats)
        return Fn.FMap(out, so if want_out_stats else None)


class Bottleneck(nn.Module):
    """conv3(conv2(conv1(x))) + shortcut(x): pre-activated 1x1x1 -> k^3 -> 1x1x1 with expansion 2, shortcut = identity or
    a FULL k-sized pre-act ConvNormAct — conv_layers.py:96-125.  Four launches of the implicit-GEMM kernel (InstanceNorm+act
    of the producer on load, statistics / residual add in the epilogue)."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=1, groups=1, dilation=1, norm="in", act="relu",
                 preact=True):
        super().__init__()
        if not preact or groups != 1 or dilation != 1:
            raise NotImplementedError("cbim_amd: Bottleneck(preact=False / groups / dilation) is not built")
        self.expansion = 2
        mid = out_ch // self.expansion
        self.conv1 = ConvNormAct(in_ch, mid, 1, stride=1, padding=0, norm=norm, act=act, preact=True)
        self.conv2 = ConvNormAct(mid, mid, kernel_size, stride=stride, norm=norm, act=act, preact=True)
        self.conv3 = ConvNormAct(mid, out_ch, 1, stride=1, padding=0, norm=norm, act=act, preact=True)
        self.shortcut = nn.Sequential()
        if in_ch != out_ch or self.conv2.strided:
            self.shortcut = ConvNormAct(in_ch, out_ch, kernel_size, stride=stride, norm=norm, act=act, preact=True)

    def forward(self, f: Fn.FMap, want_out_stats=True) -> Fn.FMap:
        if self.conv1.norm_kind == "in" and self.conv2.strided:
            f = Fn.ensure_stats(f)
            a = self.conv1.act_code
            y1, s1 = Fn.NormConvFn.apply(f.t, f.stats, self.conv1.conv.weight, a, None, True, None, IN_EPS)
            y2 = self.conv2.raw_conv(Fn.NormActFn.apply(y1, s1, a))
            s2 = Fn.ensure_stats(Fn.FMap(y2, None)).stats
            sc = self.shortcut.raw_conv(Fn.NormActFn.apply(f.t, f.stats, a))
            out, so = Fn.NormConvFn.apply(y2, s2, self.conv3.conv.weight, a, sc, want_out_stats, None, IN_EPS)
            return Fn.FMap(out, so if want_out_stats else None)
        if self.conv1.norm_kind != "in":
            res = self.shortcut.apply_generic(f.t) if isinstance(self.shortcut, ConvNormAct) else f.t
            y = self.conv2.apply_generic(self.conv1.apply_generic(f.t))
            return Fn.FMap(self.conv3.apply_generic(y, res=res), None)
        f = Fn.ensure_stats(f)
        a = self.conv1.act_code
        y1, s1 = Fn.NormConvFn.apply(f.t, f.stats, self.conv1.conv.weight, a, None, True, None, IN_EPS)
        y2, s2 = Fn.NormConvFn.apply(y1, s1, self.conv2.conv.weight, a, None, True, None, IN_EPS)
        if isinstance(self.shortcut, ConvNormAct):
            sc, _ = Fn.NormConvFn.apply(f.t, f.stats, self.shortcut.conv.weight, a, None, False, None, IN_EPS)
        else:
            sc = f.t
        out, so = Fn.NormConvFn.apply(y2, s2, self.conv3.conv.weight, a, sc, want_out_stats, None, IN_EPS)
        return Fn.FMap(out, so if want_out_stats else None)


class DepthwiseSeparableConv(nn.Module):
    """depthwise k^3 (groups=C) then pointwise 1^3, no norm/act in between — conv_layers.py:126-157.
    forward: y = pointwise(depthwise(a(IN(x)))) [+ res], a given by (stats, act) of the caller's pre-norm."""

    def __init__(self, in_ch, out_ch, stride=1, kernel_size=3, bias=False):
        super().__init__()
        if bias or stride != 1:
            raise NotImplementedError("cbim_amd: DepthwiseSeparableConv with bias/stride is not built")
        k = _k3(kernel_size)
        self.depthwise = nn.Conv3d(in_ch, in_ch, kernel_size=k, stride=1, padding=[i // 2 for i in k], groups=in_ch,
                                   bias=False)
        self.pointwise = nn.Conv3d(in_ch, out_ch, kernel_size=1, bias=False)

    def forward(self, x, stats=None, act=0, res=None, want_stats=False) -> Fn.FMap:
        t, _, _ = Fn.DWConvFn.apply(x, stats, self.depthwise.weight, act, False)
        y, so = Fn.NormConvFn.apply(t, None, self.pointwise.weight, 0, res, want_stats, None, IN_EPS)
        return Fn.FMap(y, so if want_stats else None)


class SEBlock(nn.Module):
    """squeeze (global mean) -> 1x1 conv -> act -> 1x1 conv -> sigmoid — conv_layers.py:159-175.
    Works on the [N, C] channel means (a few hundred numbers): plain torch ops; the gate is applied
    inside the next conv's fused normalisation (functional.NormConvFn `se`)."""

    def __init__(self, in_ch, ratio=4, act="relu"):
        super().__init__()
        self.squeeze = nn.Identity()
        self.excitation = nn.Sequential(nn.Conv3d(in_ch, in_ch // ratio, kernel_size=1), nn.ReLU(),
                                        nn.Conv3d(in_ch // ratio, in_ch, kernel_size=1), nn.Sigmoid())
        if act != "relu":   # the reference passes no act here, so it is always nn.ReLU (conv_layers.py:223)
            raise NotImplementedError("cbim_amd: SEBlock activation other than ReLU is not built")

    def gate(self, mean):   # mean: float32 [N, C]
        import torch
        import torch.nn.functional as F
        c1, c2 = self.excitation[0], self.excitation[2]
        if _SE_KERNELS and mean.dtype == torch.float32 and mean.dim() == 2 and mean.device == c1.weight.device:
            return Fn.SEGateFn.apply(mean, c1.weight.flatten(1), c1.bias, c2.weight.flatten(1), c2.bias)     # round 6: 2 + 3 launches
        h = F.relu(F.linear(mean, c1.weight.flatten(1), c1.bias))
        return torch.sigmoid(F.linear(h, c2.weight.flatten(1), c2.bias))


def _se_scale(t, gate):
    """x * excitation (conv_layers.py:175) on a channels-last tensor: gate [N, C] broadcast over the voxels"""
    return t * gate.to(t.dtype).view(gate.shape[0], 1, 1, 1, gate.shape[1])


class MBConv(nn.Module):
    """expand 1x1 -> depthwise k^3 -> SE -> project 1x1, every ConvNormAct pre-activated, identity
    shortcut — conv_layers.py:197-238 (the MedFormer feed-forward, medformer_utils.py:124)."""

    def __init__(self, in_ch, out_ch, expansion=4, kernel_size=3, stride=1, ratio=4, p=0, se=True, norm="in",
                 act="relu"):
        super().__init__()
        if in_ch != out_ch or stride != 1 or p or not se or expansion == 1:
            raise NotImplementedError("cbim_amd: only the MedFormer MBConv variant (in==out, stride 1, SE, p=0) is built")
        expanded = expansion * in_ch
        k = _k3(kernel_size)
        self.expand_proj = ConvNormAct(in_ch, expanded, kernel_size=1, padding=0, norm=norm, act=act, preact=True)
        self.depthwise = ConvNormAct(expanded, expanded, kernel_size=k, groups=expanded, norm=norm, act=act, preact=True)
        self.se = SEBlock(expanded, ratio=ratio)
        self.pointwise = ConvNormAct(expanded, out_ch, kernel_size=1, padding=0, norm=norm, act=None, preact=True)
        self.drop_path = nn.Identity()
        self.shortcut = nn.Sequential()

    def forward(self, f: Fn.FMap, want_out_stats=True) -> Fn.FMap:
        if self.expand_proj.norm_kind != "in":
            # `norm: bn | ln` (round 6; no shipped yaml): the composed path — every pre-activation written by one streaming pass
            # (affine norm kernels), the convolutions read it as it is, the SE gate is one broadcast multiply
            e = self.expand_proj.apply_generic(f.t)
            d, mean, _ = Fn.DWConvFn.apply(self.depthwise._norm_act(e), None, self.depthwise.conv.weight, 0, True)
            gate = self.se.gate(mean)
            return Fn.FMap(self.pointwise.apply_generic(_se_scale(d, gate), res=f.t), None)
        f = Fn.ensure_stats(f)
        act = self.expand_proj.act_code
        e, se_ = Fn.NormConvFn.apply(f.t, f.stats, self.expand_proj.conv.weight, act, None, True, None, IN_EPS)
        d, mean, ds = Fn.DWConvFn.apply(e, se_, self.depthwise.conv.weight, act, True)
        gate = self.se.gate(mean)
        y, so = Fn.NormConvFn.apply(d, ds, self.pointwise.conv.weight, 0, f.t, want_out_stats, gate, IN_EPS)
        return Fn.FMap(y, so if want_out_stats else None)


class FusedMBConv(nn.Module):
    """conv (k = 1 as MedFormer builds it) -> SE -> project 1x1, pre-activated ConvNormActs, identity shortcut —
    conv_layers.py:240-281 of the reference, the feed-forward of BidirectionAttentionBlock under proj_type 'linear'
    (medformer_utils.py:121-122).  Two launches of the row GEMM (InstanceNorm + act on load; the SE gate folded into the
    second one's normalisation) + the [N, C] excitation."""

    def __init__(self, in_ch, out_ch, expansion=4, kernel_size=3, stride=1, ratio=4, p=0, se=True, norm="in", act="relu"):
        super().__init__()
        k = _k3(kernel_size)
        if in_ch != out_ch or stride != 1 or p or not se or k != [1, 1, 1]:
            raise NotImplementedError("cbim_amd: only the MedFormer FusedMBConv variant (in==out, 1x1x1, stride 1, SE, p=0) is built")
        expanded = expansion * in_ch
        self.stride = stride
        self.conv3x3 = ConvNormAct(in_ch, expanded, kernel_size=k, padding=0, norm=norm, act=act, preact=True)
        self.se_block = SEBlock(expanded, ratio=ratio)
        self.pointwise = ConvNormAct(expanded, out_ch, kernel_size=1, padding=0, norm=norm, act=None, preact=True)
        self.drop_path = nn.Identity()
        self.shortcut = nn.Sequential()

    def forward(self, f: Fn.FMap, want_out_stats=True) -> Fn.FMap:
        if self.conv3x3.norm_kind != "in":            # `norm: bn | ln`: the composed path (see MBConv.forward)
            e = self.conv3x3.apply_generic(f.t)
            gate = self.se_block.gate(e.float().mean(dim=(1, 2, 3)))
            return Fn.FMap(self.pointwise.apply_generic(_se_scale(e, gate), res=f.t), None)
        f = Fn.ensure_stats(f)
        act = self.conv3x3.act_code
        e, se_ = Fn.NormConvFn.apply(f.t, f.stats, self.conv3x3.conv.weight, act, None, True, None, IN_EPS)
        gate = self.se_block.gate(Fn.ChannelMeanFn.apply(e, se_))
        y, so = Fn.NormConvFn.apply(e, se_, self.pointwise.conv.weight, 0, f.t, want_out_stats, gate, IN_EPS)
        return Fn.FMap(y, so if want_out_stats else None)
