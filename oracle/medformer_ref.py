"""Oracle: MedFormer forward (and, through torch autograd, backward).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

A functional restatement of ``/root/reference/model/dim3/{medformer,medformer_utils,conv_layers,
trans_layers}.py`` over a ``state_dict`` with the reference's own parameter names, written with stock
``torch`` ops in NCDHW layout on the CPU; dtype-generic (run it in float64 for a ground truth).
Pinned against ``tests/golden/medformer_tiny_32.npz`` (outputs + every gradient of the REAL reference,
``tests/golden/make_golden_medformer.py``) by ``tests/test_oracle.py``.  Line numbers are in
``/root/reference/model/dim3``.  Restated: conv_block BasicBlock, proj_type 'depthwise' (every shipped yaml) and 'linear'
(round 5, pinned by ``medformer_linear_tiny.npz``), norm 'in' and (round 6, ``medformer_bn_tiny.npz`` / ``medformer_ln_tiny.npz``)
'bn' / 'ln' — told apart by the norm parameters the state_dict holds —, dropout 0.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from .unet_ref import _act, _k3, _norm, _pad, conv_norm_act, named_norm

EPS_DEFAULT = 1e-5   # nn.InstanceNorm3d / nn.LayerNorm default (medformer_utils.py:112-113,158; trans_layers.py:38)


def dw_separable(sd, p, x, k):
    """The feature-side projection of the configured proj_type (medformer_utils.py:26-31,153-156): 'depthwise' =
    DepthwiseSeparableConv.forward (conv_layers.py:152-156): depthwise k^3 (groups=C) -> pointwise 1^3; 'linear' = one bias-free
    1x1x1 nn.Conv3d (its state_dict holds `<p>weight` instead of `<p>depthwise.weight` / `<p>pointwise.weight`)."""
    if p + "depthwise.weight" not in sd:
        return F.conv3d(x, sd[p + "weight"])
    w = sd[p + "depthwise.weight"]
    return F.conv3d(F.conv3d(x, w, None, 1, _pad(k), 1, w.shape[0]), sd[p + "pointwise.weight"])


def cna_preact(sd, p, x, k, act, groups=1):
    """pre-activation ConvNormAct (conv_layers.py:48-49), norm eps 1e-4 (:40); act=None -> no activation."""
    h = _norm(sd, p, x)
    if act:
        h = _act(h, act)
    w = sd[p + "conv.weight"]
    return F.conv3d(h, w, None, 1, [i // 2 for i in w.shape[2:]], 1, groups)


def mbconv(sd, p, x, k, act):
    """MBConv.forward (conv_layers.py:224-238) with in==out, stride 1, SE, p=0."""
    h = cna_preact(sd, p + "expand_proj.", x, 1, act)                                   # :227
    h = cna_preact(sd, p + "depthwise.", h, k, act, groups=h.shape[1])                  # :228
    s = h.mean((2, 3, 4), keepdim=True)                                                 # SEBlock :171
    s = F.conv3d(s, sd[p + "se.excitation.0.weight"], sd[p + "se.excitation.0.bias"])
    s = torch.sigmoid(F.conv3d(F.relu(s), sd[p + "se.excitation.2.weight"], sd[p + "se.excitation.2.bias"]))
    h = h * s                                                                           # :175
    h = cna_preact(sd, p + "pointwise.", h, 1, None)                                    # :232 (act=False)
    return h + x                                                                        # :236, identity shortcut


def fused_mbconv(sd, p, x, act):
    """FusedMBConv.forward (conv_layers.py:268-281) as BidirectionAttentionBlock builds it for proj_type 'linear'
    (medformer_utils.py:121-122: kernel_size 1, in == out, stride 1, SE, p = 0)."""
    h = cna_preact(sd, p + "conv3x3.", x, 1, act)                                       # :271
    s = h.mean((2, 3, 4), keepdim=True)                                                 # SEBlock :171
    s = F.conv3d(s, sd[p + "se_block.excitation.0.weight"], sd[p + "se_block.excitation.0.bias"])
    s = torch.sigmoid(F.conv3d(F.relu(s), sd[p + "se_block.excitation.2.weight"], sd[p + "se_block.excitation.2.bias"]))
    h = cna_preact(sd, p + "pointwise.", h * s, 1, None)                                # :273-275
    return h + x                                                                        # :279, identity shortcut


def _split_heads(t, heads):
    """rearrange1 (medformer_utils.py:43-51): 'b (dim_head heads) d h w -> b heads (dhw) dim_head'."""
    b, c = t.shape[:2]
    return t.reshape(b, c // heads, heads, -1).permute(0, 2, 3, 1)


def _merge_heads(t, spatial):
    """rearrange2 (medformer_utils.py:52-59): 'b heads l dim_head -> b (dim_head heads) d h w'."""
    b, heads, l, dh = t.shape
    return t.permute(0, 3, 1, 2).reshape(b, heads * dh, *spatial)


def bidirection_attention(sd, p, feat, smap, heads, k, no_map_out):
    """BidirectionAttention.forward (medformer_utils.py:63-97)."""
    fq, fv = dw_separable(sd, p + "feat_qv.", feat, k).chunk(2, dim=1)                  # :67
    mq, mv = F.conv3d(smap, sd[p + "map_qv.weight"]).chunk(2, dim=1)                    # :68
    dh = fq.shape[1] // heads
    fq, fv, mq, mv = (_split_heads(t, heads) for t in (fq, fv, mq, mv))
    attn = torch.einsum("bhid,bhjd->bhij", fq, mq) * dh ** -0.5                         # :77-78
    feat_out = torch.einsum("bhij,bhjd->bhid", F.softmax(attn, dim=-1), mv)             # :80,85
    map_out = torch.einsum("bhji,bhjd->bhid", F.softmax(attn, dim=-2), fv)              # :82,90
    feat_out = dw_separable(sd, p + "feat_out.", _merge_heads(feat_out, feat.shape[2:]), k)   # :87,95
    map_out = _merge_heads(map_out, smap.shape[2:])
    if not no_map_out:
        map_out = F.conv3d(map_out, sd[p + "map_out.weight"])                           # :96
    return feat_out, map_out


def attention_block(sd, p, x, smap, heads, k, act, no_map_out):
    """BidirectionAttentionBlock.forward (medformer_utils.py:126-138)."""
    feat = named_norm(sd, p + "norm1.", x, EPS_DEFAULT)                                 # norm1 :128 (built with the default eps, :112)
    mapp = named_norm(sd, p + "norm2.", smap, EPS_DEFAULT)                              # norm2 :129
    out, mapp = bidirection_attention(sd, p + "attn.", feat, mapp, heads, k, no_map_out)
    if (p + "shortcut.conv.weight") in sd:                                              # :119-121
        out = out + cna_preact(sd, p + "shortcut.", x, 1, act)
    else:
        out = out + x
    if (p + "feedforward.conv3x3.conv.weight") in sd:                                  # proj_type 'linear': FusedMBConv (:121-122)
        return fused_mbconv(sd, p + "feedforward.", out, act), mapp + smap
    return mbconv(sd, p + "feedforward.", out, k, act), mapp + smap                     # :134-136


def patch_merging(sd, p, x, scale, k):
    """PatchMerging.forward (medformer_utils.py:159-175)."""
    parts = [x[:, :, i::scale[0], j::scale[1], kk::scale[2]]
             for i in range(scale[0]) for j in range(scale[1]) for kk in range(scale[2])]
    return dw_separable(sd, p + "reduction.", named_norm(sd, p + "norm.", torch.cat(parts, 1), EPS_DEFAULT), k)    # :158,172


def semantic_map_generation(sd, p, x, map_size):
    """SemanticMapGeneration.forward (medformer_utils.py:215-228)."""
    B = x.shape[0]
    feat = F.conv3d(x, sd[p + "base_proj.weight"], None, 1, 1)
    wm = F.conv3d(x, sd[p + "semantic_proj.weight"], None, 1, 1)
    wm = F.softmax(wm.reshape(B, wm.shape[1], -1), dim=2)
    return torch.einsum("bij,bkj->bik", feat.reshape(B, feat.shape[1], -1), wm).reshape(B, feat.shape[1], *map_size)


def _vit_attention(sd, p, x, heads):
    """trans_layers.Attention.forward (:74-95), '(heads dim_head)' split."""
    B, L, _ = x.shape
    q, k, v = (t.reshape(B, L, heads, -1).permute(0, 2, 1, 3) for t in F.linear(x, sd[p + "to_qkv.weight"]).chunk(3, -1))
    attn = F.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * q.shape[-1] ** -0.5, dim=-1)
    o = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(B, L, -1)
    return F.linear(o, sd[p + "to_out.weight"], sd[p + "to_out.bias"])


def semantic_map_fusion(sd, p, maps, heads, depth):
    """SemanticMapFusion.forward (medformer_utils.py:250-261) + TransformerBlock (trans_layers.py:98-118)."""
    B, _, D, H, W = maps[0].shape
    toks = [F.conv3d(m, sd[f"{p}in_proj.{i}.weight"]).flatten(2).permute(0, 2, 1) for i, m in enumerate(maps)]
    x = torch.cat(toks, dim=1)
    dim = x.shape[-1]
    for l in range(depth):
        q = f"{p}fusion.layers.{l}."
        h = F.layer_norm(x, (dim,), sd[q + "0.norm.weight"], sd[q + "0.norm.bias"], EPS_DEFAULT)
        x = _vit_attention(sd, q + "0.fn.", h, heads) + x
        h = F.layer_norm(x, (dim,), sd[q + "1.norm.weight"], sd[q + "1.norm.bias"], EPS_DEFAULT)
        h = F.linear(F.gelu(F.linear(h, sd[q + "1.fn.fc1.weight"], sd[q + "1.fn.fc1.bias"])),
                     sd[q + "1.fn.fc2.weight"], sd[q + "1.fn.fc2.bias"])
        x = h + x
    outs = x.chunk(len(maps), dim=1)
    return [F.conv3d(o.permute(0, 2, 1).reshape(B, dim, D, H, W), sd[f"{p}out_proj.{i}.weight"])
            for i, o in enumerate(outs)]


def _count(sd, prefix):
    n = 0
    while any(k.startswith(f"{prefix}{n}.") for k in sd):
        n += 1
    return n


def _down(sd, p, x, scale, k, heads, map_size, act, map_generate):
    """down_block.forward (medformer_utils.py:306-319)."""
    x = patch_merging(sd, p + "patch_merging.", x, scale, k)
    for i in range(_count(sd, p + "conv_blocks.")):
        x = basic_block_act(sd, f"{p}conv_blocks.{i}.", x, k, act)
    smap = semantic_map_generation(sd, p + "map_gen.", x, map_size) if map_generate else None
    for i in range(_count(sd, p + "trans_blocks.blocks.")):
        x, smap = attention_block(sd, f"{p}trans_blocks.blocks.{i}.", x, smap, heads, k, act, False)
    return x, smap


def _up(sd, p, x1, x2, map1, map2, k, heads, act, map_shortcut, no_map_out):
    """up_block.forward (medformer_utils.py:352-372): cat([upsampled, skip])."""
    x1 = F.interpolate(x1, size=x2.shape[-3:], mode="trilinear", align_corners=True)
    feat = torch.cat([x1, x2], dim=1)
    if map_shortcut and map2 is not None:
        smap = F.conv3d(torch.cat([map1, map2], dim=1), sd[p + "map_reduction.weight"])
    else:
        smap = map1
    nb = _count(sd, p + "trans_blocks.blocks.")
    for i in range(nb):
        feat, smap = attention_block(sd, f"{p}trans_blocks.blocks.{i}.", feat, smap, heads, k, act,
                                     no_map_out and i == nb - 1)
    for i in range(_count(sd, p + "conv_blocks.")):
        feat = basic_block_act(sd, f"{p}conv_blocks.{i}.", feat, k, act)
    return feat, smap


def basic_block_act(sd, p, x, k, act):
    """BasicBlock.forward (conv_layers.py:86-94) with the configured activation."""
    h = conv_norm_act(sd, p + "conv1.", x, k, preact=True, act=act)
    h = conv_norm_act(sd, p + "conv2.", h, k, preact=True, act=act)
    if (p + "shortcut.conv.weight") in sd:
        x = conv_norm_act(sd, p + "shortcut.", x, k, preact=True, act=act)
    return h + x


def medformer_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, *, map_size, num_heads, fusion_heads,
                      fusion_depth, kernel_size, scale, act="relu", aux_loss=False) -> List[torch.Tensor]:
    """MedFormer.forward (medformer.py:73-101)."""
    ks = [_k3(k) for k in kernel_size]
    sc = [_k3(s) for s in scale]
    x0 = F.conv3d(x, sd["inc.conv1.weight"], None, 1, _pad(ks[0]))                      # inconv :272-276
    x0 = basic_block_act(sd, "inc.conv2.", x0, ks[0], act)
    x1, _ = _down(sd, "down1.", x0, sc[0], ks[1], 1, map_size, act, False)
    x2, map2 = _down(sd, "down2.", x1, sc[1], ks[2], num_heads[1], map_size, act, True)
    x3, map3 = _down(sd, "down3.", x2, sc[2], ks[3], num_heads[2], map_size, act, True)
    x4, map4 = _down(sd, "down4.", x3, sc[3], ks[4], num_heads[3], map_size, act, True)
    maps = semantic_map_fusion(sd, "map_fusion.", [map2, map3, map4], fusion_heads, fusion_depth)
    out, smap = _up(sd, "up1.", x4, x3, maps[2], maps[1], ks[3], num_heads[4], act, True, False)
    out, smap = _up(sd, "up2.", out, x2, smap, maps[0], ks[2], num_heads[5], act, True, True)
    aux = None
    if aux_loss:
        aux = F.conv3d(out, sd["aux_out.weight"], sd["aux_out.bias"])
        aux = F.interpolate(aux, size=x.shape[-3:], mode="trilinear", align_corners=True)
    out, smap = _up(sd, "up3.", out, x1, smap, None, ks[1], num_heads[6], act, False, False)
    out, smap = _up(sd, "up4.", out, x0, smap, None, ks[0], num_heads[7], act, False, False)
    out = F.conv3d(out, sd["outc.weight"], sd["outc.bias"])
    return [out, aux] if aux_loss else [out]
