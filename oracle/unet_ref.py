"""Oracle: 3D UNet / ResUNet forward (and, through torch autograd, backward).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

A functional restatement of ``/root/reference/model/dim3/{unet,unet_utils,
conv_layers,utils}.py`` over a ``state_dict`` that uses the reference's own
parameter names, written with stock ``torch.nn.functional`` ops in NCDHW layout
on the CPU.  All line numbers below are in ``/root/reference``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

IN_EPS_CONV = 1e-4  # conv_layers.py:40,42  norm(ch, eps=1e-4)


def _k3(k):
    return [k] * 3 if isinstance(k, int) else list(k)


def _pad(k):  # conv_layers.py:62,77 / unet_utils.py:13  pad_size = [i//2 for i in kernel_size]
    return [i // 2 for i in k]


def _act(x, act: str = "relu"):
    # model/dim3/utils.py:23-30 act map; UNet passes no act -> nn.ReLU (conv_layers.py:23)
    if act == "relu":
        return F.relu(x)
    if act == "lrelu":
        return F.leaky_relu(x, 0.01)
    if act == "gelu":
        return F.gelu(x)
    if act == "swish":
        return F.silu(x)
    raise ValueError(act)


def instance_norm(x, eps=IN_EPS_CONV):
    """nn.InstanceNorm3d(C, eps) with affine=False, no running stats
    (model/dim3/utils.py:15-21, conv_layers.py:40-42): per (n, c) biased variance."""
    return F.instance_norm(x, None, None, None, None, True, 0.0, eps)


BN_MOMENTUM = 0.1   # nn.BatchNorm3d default (model/dim3/utils.py:16: norm_map['bn'] = nn.BatchNorm3d, built with eps=1e-4 only)


def named_norm(sd, key, x, eps, training=True):
    """A norm module of the configured kind under the state_dict prefix `key` (e.g. "down2.patch_merging.norm."):
    InstanceNorm3d(eps, affine=False) when the state_dict holds no parameters there (`norm: in`), the channels-first LayerNorm of
    trans_layers.py:120-149 when it holds weight / bias only (`norm: ln`), else nn.BatchNorm3d (`norm: bn`): batch statistics +
    in-place running-statistics update in training mode, the running statistics in eval mode (F.batch_norm semantics;
    num_batches_tracked counted)."""
    if key + "weight" not in sd:
        return instance_norm(x, eps)
    if key + "running_mean" not in sd:
        u = x.mean(1, keepdim=True)
        v = (x - u).pow(2).mean(1, keepdim=True)
        shape = (1, -1) + (1,) * (x.dim() - 2)
        return sd[key + "weight"].view(shape) * ((x - u) / torch.sqrt(v + eps)) + sd[key + "bias"].view(shape)
    rm, rv = sd[key + "running_mean"], sd[key + "running_var"]
    if training and key + "num_batches_tracked" in sd:
        sd[key + "num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[key + "weight"], sd[key + "bias"], training, BN_MOMENTUM, eps)


def _norm(sd, prefix, x, training=True):
    """self.norm of a ConvNormAct (conv_layers.py:40-43): built with eps=1e-4 whatever the kind."""
    return named_norm(sd, prefix + "norm.", x, IN_EPS_CONV, training)


def conv_norm_act(sd, prefix, x, k, preact, act="relu", training=True, stride=1):
    """ConvNormAct.forward (conv_layers.py:46-53); conv has bias=False (:23); `stride` as given to nn.Conv3d (:29-38)."""
    w = sd[prefix + "conv.weight"]
    b = sd.get(prefix + "conv.bias")
    if preact:  # :48-49  conv(act(norm(x)))
        return F.conv3d(_act(_norm(sd, prefix, x, training), act), w, b, stride, _pad(k))
    return _act(_norm(sd, prefix, F.conv3d(x, w, b, stride, _pad(k)), training), act)  # :51


def single_conv(sd, prefix, x, k, training=True, stride=1):
    """SingleConv.forward (conv_layers.py:56-68): one post-activation ConvNormAct."""
    return conv_norm_act(sd, prefix + "conv.", x, k, preact=False, training=training, stride=stride)


def basic_block(sd, prefix, x, k, training=True, stride=1):
    """BasicBlock.forward (conv_layers.py:86-94), preact=True default (:72).
    shortcut is a full k-sized pre-act ConvNormAct when in_ch != out_ch (:83-84)."""
    out = conv_norm_act(sd, prefix + "conv1.", x, k, preact=True, training=training, stride=stride)   # :79 stride on conv1 ...
    out = conv_norm_act(sd, prefix + "conv2.", out, k, preact=True, training=training)
    if prefix + "shortcut.conv.weight" in sd:
        res = conv_norm_act(sd, prefix + "shortcut.", x, k, preact=True, training=training, stride=stride)   # ... and the shortcut (:83-84)
    else:
        res = x
    return out + res  # :92


def bottleneck(sd, prefix, x, k, training=True, stride=1):
    """Bottleneck.forward (conv_layers.py:116-125): 1x1 -> kxk -> 1x1, all pre-act."""
    out = conv_norm_act(sd, prefix + "conv1.", x, [1, 1, 1], preact=True, training=training)
    out = conv_norm_act(sd, prefix + "conv2.", out, k, preact=True, training=training, stride=stride)   # :108 stride on the k^3 conv
    out = conv_norm_act(sd, prefix + "conv3.", out, [1, 1, 1], preact=True, training=training)
    if prefix + "shortcut.conv.weight" in sd:
        res = conv_norm_act(sd, prefix + "shortcut.", x, k, preact=True, training=training, stride=stride)
    else:
        res = x
    return out + res


_BLOCKS = {"SingleConv": single_conv, "BasicBlock": basic_block, "Bottleneck": bottleneck}


def unet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, *, scale, kernel_size,
                 block: str = "BasicBlock", return_features: bool = False, training: bool = True, pool: bool = True):
    """UNet.forward (unet.py:50-64).

    inconv      unet_utils.py:18-21   raw Conv3d (bias False) then one block
    down_block  unet_utils.py:35-46   MaxPool3d(scale) -> block -> block
    up_block    unet_utils.py:68-75   trilinear(align_corners=True) to skip size ->
                                      cat([skip, up]) -> block -> block
    outc        unet.py:47            1x1x1 conv with bias
    """
    from functools import partial
    blk = partial(_BLOCKS[block], training=training)   # (only `norm: bn` state_dicts depend on the mode)
    ks = [_k3(k) for k in kernel_size]
    sc = [_k3(s) for s in scale]
    feats = OrderedDict()

    h = F.conv3d(x, sd["inc.conv1.weight"], None, 1, _pad(ks[0]))
    x1 = blk(sd, "inc.conv2.", h, ks[0])
    feats["x1"] = x1
    skips = [x1]
    cur = x1
    for lvl in range(4):  # down1..down4 use kernel_size[lvl+1], scale[lvl]  (unet.py:37-40)
        if pool:   # unet_utils.py:35-37  MaxPool3d -> block -> block  (Sequential indices 1, 2)
            cur = F.max_pool3d(cur, sc[lvl])
            cur = blk(sd, f"down{lvl+1}.conv.1.", cur, ks[lvl + 1])
            cur = blk(sd, f"down{lvl+1}.conv.2.", cur, ks[lvl + 1])
        else:      # unet_utils.py:38-39  block(stride=down_scale) -> block  (Sequential indices 0, 1)
            cur = blk(sd, f"down{lvl+1}.conv.0.", cur, ks[lvl + 1], stride=tuple(sc[lvl]))
            cur = blk(sd, f"down{lvl+1}.conv.1.", cur, ks[lvl + 1])
        feats[f"x{lvl+2}"] = cur
        skips.append(cur)
    out = skips[4]
    for i in range(4):  # up1..up4 use kernel_size[3-i]  (unet.py:42-45)
        skip = skips[3 - i]
        up = F.interpolate(out, size=skip.shape[2:], mode="trilinear", align_corners=True)
        out = torch.cat([skip, up], dim=1)  # unet_utils.py:71
        out = blk(sd, f"up{i+1}.conv.0.", out, ks[3 - i])
        out = blk(sd, f"up{i+1}.conv.1.", out, ks[3 - i])
        feats[f"u{i+1}"] = out
    logits = F.conv3d(out, sd["outc.weight"], sd["outc.bias"])
    if return_features:
        return logits, feats
    return logits


# ----------------------------------------------------------------------------------------
# Deterministic "reference default init" state_dict (so that tests on the GPU box, where
# /root/reference does not exist, can rebuild exactly the weights the reference constructor
# would draw for a given torch seed).  Creation ORDER mirrors unet.py:35-47 /
# unet_utils.py / conv_layers.py so the RNG stream is consumed identically; nn.Conv3d's own
# reset_parameters is used (kaiming_uniform(a=sqrt(5)) + uniform bias).
# ----------------------------------------------------------------------------------------

def _bn(sd, name, ch):
    """nn.BatchNorm3d(ch) entries in state_dict order (its reset_parameters draws no random numbers)"""
    sd[name + "weight"] = torch.ones(ch)
    sd[name + "bias"] = torch.zeros(ch)
    sd[name + "running_mean"] = torch.zeros(ch)
    sd[name + "running_var"] = torch.ones(ch)
    sd[name + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _cna(sd, prefix, cin, cout, k, norm, preact):
    """a ConvNormAct's entries (conv_layers.py:16-43): conv first, then the norm over in_ch (preact) or out_ch"""
    _conv(sd, prefix + "conv.", cin, cout, k)
    if norm == "bn":
        _bn(sd, prefix + "norm.", cin if preact else cout)
    elif norm == "ln":       # LayerNorm(ch): weight = ones, bias = zeros (trans_layers.py:130-131), no random draws
        ch = cin if preact else cout
        sd[prefix + "norm.weight"], sd[prefix + "norm.bias"] = torch.ones(ch), torch.zeros(ch)


def _conv(sd, name, cin, cout, k, bias=False):
    m = torch.nn.Conv3d(cin, cout, kernel_size=k, padding=_pad(k), bias=bias)
    sd[name + "weight"] = m.weight.detach().clone()
    if bias:
        sd[name + "bias"] = m.bias.detach().clone()


def _make_block(sd, prefix, block, cin, cout, k, norm="in", strided=False):
    if block == "SingleConv":
        _cna(sd, prefix + "conv.", cin, cout, k, norm, False)
    elif block == "BasicBlock":  # conv_layers.py:79-84 creation order conv1, conv2, shortcut
        _cna(sd, prefix + "conv1.", cin, cout, k, norm, True)
        _cna(sd, prefix + "conv2.", cout, cout, k, norm, True)
        if cin != cout or strided:   # conv_layers.py:83  `if stride != 1 or in_ch != out_ch`
            _cna(sd, prefix + "shortcut.", cin, cout, k, norm, True)
    elif block == "Bottleneck":  # conv_layers.py:106-113
        _cna(sd, prefix + "conv1.", cin, cout // 2, [1, 1, 1], norm, True)
        _cna(sd, prefix + "conv2.", cout // 2, cout // 2, k, norm, True)
        _cna(sd, prefix + "conv3.", cout // 2, cout, [1, 1, 1], norm, True)
        if cin != cout or strided:
            _cna(sd, prefix + "shortcut.", cin, cout, k, norm, True)
    else:
        raise KeyError(block)


def make_unet_state_dict(in_ch, base_ch, num_classes, kernel_size, block="BasicBlock", seed=None, norm="in", pool=True):
    if seed is not None:
        torch.manual_seed(seed)
    ks = [_k3(k) for k in kernel_size]
    sd = OrderedDict()
    b = base_ch
    _conv(sd, "inc.conv1.", in_ch, b, ks[0])
    _make_block(sd, "inc.conv2.", block, b, b, ks[0], norm)
    chans = [b, 2 * b, 4 * b, 8 * b, 10 * b]  # unet.py:37-40
    for lvl in range(4):
        i0 = 1 if pool else 0      # nn.Sequential index of the level's first block (a MaxPool3d occupies 0 when pooling)
        _make_block(sd, f"down{lvl+1}.conv.{i0}.", block, chans[lvl], chans[lvl + 1], ks[lvl + 1], norm, strided=not pool)
        _make_block(sd, f"down{lvl+1}.conv.{i0 + 1}.", block, chans[lvl + 1], chans[lvl + 1], ks[lvl + 1], norm)
    for i in range(4):  # up1: (10b -> 8b) ... up4: (2b -> b); block in = in+out (unet_utils.py:62)
        cin, cout = chans[4 - i], chans[3 - i]
        _make_block(sd, f"up{i+1}.conv.0.", block, cin + cout, cout, ks[3 - i], norm)
        _make_block(sd, f"up{i+1}.conv.1.", block, cout, cout, ks[3 - i], norm)
    _conv(sd, "outc.", b, num_classes, [1, 1, 1], bias=True)
    return sd


def state_dict_checksum(sd) -> float:
    """Order-sensitive scalar fingerprint of a state_dict (float64)."""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        v = v.double().flatten()
        w = torch.arange(1, v.numel() + 1, dtype=torch.float64) / v.numel()
        acc += (i + 1) * float((v * w).sum())
    return acc


def unetpp_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, *, scale, kernel_size, block="BasicBlock", training=True) -> torch.Tensor:
    """UNetPlusPlus.forward (/root/reference/model/dim3/unetpp.py:52-76); layers are nn.Sequential of two blocks
    (make_layer :79-88), pooling nn.MaxPool3d(scale[i]), upsampling nn.Upsample(scale_factor, trilinear,
    align_corners=True)."""
    blk = _BLOCKS[block]
    ks = [_k3(k) for k in kernel_size]
    sc = [tuple(_k3(s)) for s in scale]

    def layer(name, t, lvl):
        t = blk(sd, f"{name}.0.", t, ks[lvl], training=training)
        return blk(sd, f"{name}.1.", t, ks[lvl], training=training)

    def up(t, i):
        return F.interpolate(t, scale_factor=tuple(float(v) for v in sc[i]), mode="trilinear", align_corners=True)

    x0_0 = layer("conv0_0", x, 0)
    x1_0 = layer("conv1_0", F.max_pool3d(x0_0, sc[0]), 1)
    x0_1 = layer("conv0_1", torch.cat([x0_0, up(x1_0, 0)], 1), 0)
    x2_0 = layer("conv2_0", F.max_pool3d(x1_0, sc[1]), 2)
    x1_1 = layer("conv1_1", torch.cat([x1_0, up(x2_0, 1)], 1), 1)
    x0_2 = layer("conv0_2", torch.cat([x0_0, x0_1, up(x1_1, 0)], 1), 0)
    x3_0 = layer("conv3_0", F.max_pool3d(x2_0, sc[2]), 3)
    x2_1 = layer("conv2_1", torch.cat([x2_0, up(x3_0, 2)], 1), 2)
    x1_2 = layer("conv1_2", torch.cat([x1_0, x1_1, up(x2_1, 1)], 1), 1)
    x0_3 = layer("conv0_3", torch.cat([x0_0, x0_1, x0_2, up(x1_2, 0)], 1), 0)
    x4_0 = layer("conv4_0", F.max_pool3d(x3_0, sc[3]), 4)
    x3_1 = layer("conv3_1", torch.cat([x3_0, up(x4_0, 3)], 1), 3)
    x2_2 = layer("conv2_2", torch.cat([x2_0, x2_1, up(x3_1, 2)], 1), 2)
    x1_3 = layer("conv1_3", torch.cat([x1_0, x1_1, x1_2, up(x2_2, 1)], 1), 1)
    x0_4 = layer("conv0_4", torch.cat([x0_0, x0_1, x0_2, x0_3, up(x1_3, 0)], 1), 0)
    return F.conv3d(x0_4, sd["output.weight"], sd["output.bias"])


def attention_unet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, *, scale, kernel_size, block="BasicBlock",
                           training=True) -> torch.Tensor:
    """AttentionUNet.forward (/root/reference/model/dim3/attention_unet.py:30-45) with attention_up_block /
    AttentionBlock (attention_unet_utils.py:6-66): nn.InstanceNorm3d default eps 1e-5 inside the gate."""
    from functools import partial
    blk = partial(_BLOCKS[block], training=training)      # (`norm: bn` state_dicts depend on the mode; the gates do not)
    ks = [_k3(k) for k in kernel_size]
    sc = [_k3(s) for s in scale]

    def in5(t):
        return F.instance_norm(t, eps=1e-5)

    def up(p, lvl, low, skip):
        x1 = F.interpolate(low, size=skip.shape[2:], mode="trilinear", align_corners=True)
        g1 = in5(F.conv3d(x1, sd[p + "attn.W_g.0.weight"]))
        s1 = in5(F.conv3d(skip, sd[p + "attn.W_x.0.weight"]))
        psi = torch.sigmoid(in5(F.conv3d(F.relu(g1 + s1), sd[p + "attn.psi.0.weight"])))
        out = torch.cat([skip * psi, x1], dim=1)
        out = blk(sd, p + "conv.0.", out, ks[lvl])
        return blk(sd, p + "conv.1.", out, ks[lvl])

    x1 = F.conv3d(x, sd["inc.conv1.weight"], None, 1, _pad(ks[0]))
    x1 = blk(sd, "inc.conv2.", x1, ks[0])
    feats = [x1]
    t = x1
    for i in range(4):
        t = F.max_pool3d(t, sc[i])
        t = blk(sd, f"down{i + 1}.conv.1.", t, ks[i + 1])
        t = blk(sd, f"down{i + 1}.conv.2.", t, ks[i + 1])
        feats.append(t)
    out = up("up1.", 3, feats[4], feats[3])
    out = up("up2.", 2, out, feats[2])
    out = up("up3.", 1, out, feats[1])
    out = up("up4.", 0, out, feats[0])
    return F.conv3d(out, sd["outc.weight"], sd["outc.bias"])
