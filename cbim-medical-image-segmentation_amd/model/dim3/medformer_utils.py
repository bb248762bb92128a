"""MedFormer building blocks with the reference's module / parameter names
(/root/reference/model/dim3/medformer_utils.py), executed by the gfx950 kernels behind
``cbim_amd.functional``.

Feature maps (L = D*H*W voxels) are channels-last ``Fn.FMap``s handled only by HIP kernels.  Semantic
maps are ``[B, C, m0, m1, m2]`` float32 torch tensors with m0*m1*m2 <= 64 positions: their 1x1
projections / InstanceNorm / residuals are a few KFLOP each and stay ordinary torch ops (autograd
differentiates them); everything that touches L voxels is a kernel.
"""
import os

import torch
import torch.nn as nn

from ... import functional as Fn
from ...ops import ACT, IN_EPS, MAP_MAX_POSITIONS
from .conv_layers import BasicBlock, ConvNormAct, DepthwiseSeparableConv, FusedMBConv, MBConv, _k3, make_norm, norm_rows
from .trans_layers import TransformerBlock

_EPS_DEFAULT = 1e-5   # nn.InstanceNorm3d default, used by norm1/norm2 and PatchMerging.norm (:112-113,:158)
# the map branch of a block on k_map_gemm (round 6); CBIM_MAP_KERNELS=0: the torch ops of rounds 1-5 (same-box A/B runs)
_MAP_KERNELS = os.environ.get("CBIM_MAP_KERNELS", "1") != "0"


def pointwise(conv: nn.Conv3d, x):
    """1x1x1 nn.Conv3d on a semantic map [B, C, m0, m1, m2] as one matmul over the channel axis (torch's conv3d
    would pick a generic MIOpen kernel that costs ~0.3 ms for these 64-position tensors)."""
    if x.shape[0] == 1:   # (2-D x 3-D matmul folds the batch through transposed clones: one mm, no copy, for one image)
        y = torch.mm(conv.weight.flatten(1), x.flatten(2).squeeze(0)).unsqueeze(0)
    else:
        y = torch.matmul(conv.weight.flatten(1), x.flatten(2))
    if conv.bias is not None:
        y = y + conv.bias[None, :, None]
    return y.reshape(x.shape[0], -1, *x.shape[2:])


def map_instance_norm(x, eps=_EPS_DEFAULT):
    """nn.InstanceNorm3d (affine=False) over the <= 64 map positions."""
    # = layer normalisation of each (image, channel) row over its positions, biased variance, no affine: ONE kernel forward
    # and one backward (var_mean / sub / rsqrt / mul were 5 launches forward and 12 backward per block, 16 blocks)
    f = x.flatten(2)
    return torch.nn.functional.layer_norm(f, (f.shape[2],), None, None, eps).reshape(x.shape)


class BidirectionAttention(nn.Module):
    """medformer_utils.py:11-97 (proj_type 'depthwise')."""

    def __init__(self, feat_dim, map_dim, out_dim, heads=4, dim_head=64, attn_drop=0., proj_drop=0.,
                 map_size=(8, 8, 8), proj_type="depthwise", kernel_size=(3, 3, 3), no_map_out=False):
        super().__init__()
        if proj_type not in ("linear", "depthwise"):
            raise ValueError(proj_type)
        if attn_drop or proj_drop:
            raise NotImplementedError("cbim_amd: attention dropout is not built (0 in every shipped config)")
        self.inner_dim, self.heads, self.dim_head = dim_head * heads, heads, dim_head
        self.scale = dim_head ** (-0.5)
        self.linear = proj_type == "linear"
        if self.linear:   # medformer_utils.py:26-28: bias-free 1x1x1 feature projections
            self.feat_qv = nn.Conv3d(feat_dim, self.inner_dim * 2, kernel_size=1, stride=1, padding=0, bias=False)
            self.feat_out = nn.Conv3d(self.inner_dim, out_dim, kernel_size=1, stride=1, padding=0, bias=False)
        else:
            self.feat_qv = DepthwiseSeparableConv(feat_dim, self.inner_dim * 2, kernel_size=kernel_size)
            self.feat_out = DepthwiseSeparableConv(self.inner_dim, out_dim, kernel_size=kernel_size)
        self.map_qv = nn.Conv3d(map_dim, self.inner_dim * 2, kernel_size=1, bias=False)
        self.map_out = nn.Identity() if no_map_out else nn.Conv3d(self.inner_dim, map_dim, kernel_size=1, bias=False)

    def forward(self, x, x_stats, mapp, res, want_stats, raw_map=None):
        """x (raw) with InstanceNorm stats (eps 1e-5) fused into the depthwise load; mapp already normalised — or (round 6)
        raw_map given: the block's un-normalised semantic map, whose norm2, map_qv, map_out and `+ semantic_map` run on the
        engine's small-GEMM kernel (functional.MapQVFn / MapOutFn); the second return value is then the block's NEW map.
        Returns (feat_out + res as FMap, map_out [B, *, m...])."""
        if self.linear:   # IN (the block's norm1) on load of the row GEMM, no activation
            qv = Fn.NormConvFn.apply(x, x_stats, self.feat_qv.weight, 0, None, False, None, IN_EPS)[0]
        else:
            qv = self.feat_qv(x, x_stats, 0).t
        if raw_map is not None:
            B, ms = raw_map.shape[0], tuple(raw_map.shape[2:])
            sm = raw_map.flatten(2)
            mq, mv = Fn.MapQVFn.apply(sm, self.map_qv.weight.flatten(1), _EPS_DEFAULT)
            fo, mo = Fn.BidirAttnFn.apply(qv, mq, mv, self.heads, self.scale)
            if self.linear:
                y, so = Fn.NormConvFn.apply(fo, None, self.feat_out.weight, 0, res, want_stats, None, IN_EPS)
                out = Fn.FMap(y, so if want_stats else None)
            else:
                out = self.feat_out(fo, None, 0, res=res, want_stats=want_stats)
            if isinstance(self.map_out, nn.Identity):
                new_map = mo.transpose(1, 2) + sm
            else:
                new_map = Fn.MapOutFn.apply(mo, self.map_out.weight.flatten(1), sm)
            return out, new_map.reshape(B, -1, *ms)
        B, ms = mapp.shape[0], tuple(mapp.shape[2:])
        mqv = pointwise(self.map_qv, mapp).flatten(2).transpose(1, 2)   # [B, M, 2*inner]
        mq, mv = mqv[..., :self.inner_dim], mqv[..., self.inner_dim:]
        fo, mo = Fn.BidirAttnFn.apply(qv, mq, mv, self.heads, self.scale)
        if self.linear:
            y, so = Fn.NormConvFn.apply(fo, None, self.feat_out.weight, 0, res, want_stats, None, IN_EPS)
            out = Fn.FMap(y, so if want_stats else None)
        else:
            out = self.feat_out(fo, None, 0, res=res, want_stats=want_stats)
        mo = mo.transpose(1, 2).reshape(B, self.inner_dim, *ms)
        return out, (mo if isinstance(self.map_out, nn.Identity) else pointwise(self.map_out, mo))


class BidirectionAttentionBlock(nn.Module):
    """medformer_utils.py:102-138."""

    def __init__(self, feat_dim, map_dim, out_dim, heads, dim_head, norm="in", act="relu", expansion=4,
                 attn_drop=0., proj_drop=0., map_size=(8, 8, 8), proj_type="depthwise", kernel_size=(3, 3, 3),
                 no_map_out=False):
        super().__init__()
        # `in`: InstanceNorm3d(feat_dim), eps 1e-5, fused into feat_qv's depthwise load / InstanceNorm3d(map_dim) inside the map
        # kernels (no parameters: nn.Identity holders).  `bn | ln` (round 6): the reference's parameter holders, composed path
        self.norm_kind = norm if norm in ("bn", "ln") else "in"
        self.norm1 = make_norm(self.norm_kind, feat_dim)
        self.norm2 = make_norm(self.norm_kind, map_dim)
        self.attn = BidirectionAttention(feat_dim, map_dim, out_dim, heads, dim_head, attn_drop=attn_drop,
                                         proj_drop=proj_drop, map_size=map_size, proj_type=proj_type,
                                         kernel_size=kernel_size, no_map_out=no_map_out)
        self.shortcut = nn.Sequential()
        if feat_dim != out_dim:
            self.shortcut = ConvNormAct(feat_dim, out_dim, 1, padding=0, norm=norm, act=act, preact=True)
        if proj_type == "linear":   # medformer_utils.py:121-122
            self.feedforward = FusedMBConv(out_dim, out_dim, expansion=expansion, kernel_size=1, act=act, norm=norm)
        else:
            self.feedforward = MBConv(out_dim, out_dim, expansion=expansion, kernel_size=kernel_size, act=act, norm=norm)

    def forward(self, f: Fn.FMap, semantic_map, want_out_stats=True):
        if self.norm_kind != "in":
            # medformer_utils.py:126-138 step by step: norm1 of the features by one streaming pass (affine norm kernels), norm2 of
            # the <= 128-position map through the holder's own forward, the attention on the normalised tensors as they are
            feat = norm_rows(self.norm1, f.t)
            mapp = self.norm2(semantic_map)
            res = self.shortcut.apply_generic(f.t) if isinstance(self.shortcut, ConvNormAct) else f.t
            out, mapp = self.attn(feat, None, mapp, res, False)
            return self.feedforward(out, want_out_stats), mapp + semantic_map
        f = Fn.ensure_stats(f)                                   # eps 1e-4 (ConvNormAct convention)
        s5 = Fn.restat(f.stats, IN_EPS, _EPS_DEFAULT)
        if isinstance(self.shortcut, ConvNormAct):
            res, _ = Fn.NormConvFn.apply(f.t, f.stats, self.shortcut.conv.weight, self.shortcut.act_code, None, False,
                                         None, IN_EPS)
        else:
            res = f.t
        M = semantic_map.shape[2] * semantic_map.shape[3] * semantic_map.shape[4]
        if _MAP_KERNELS and semantic_map.dtype == torch.float32 and M <= MAP_MAX_POSITIONS and semantic_map.device == f.t.device:
            # round 6: norm2 -> map_qv -> (attention) -> map_out -> + semantic_map on the engine's own small-GEMM kernel
            out, new_map = self.attn(f.t, s5, None, res, True, raw_map=semantic_map)
            return self.feedforward(out, want_out_stats), new_map
        mapp = map_instance_norm(semantic_map)
        out, mapp = self.attn(f.t, s5, mapp, res, True)
        out = self.feedforward(out, want_out_stats)
        return out, mapp + semantic_map


class PatchMerging(nn.Module):
    """medformer_utils.py:140-175 (proj_type 'depthwise')."""

    def __init__(self, dim, out_dim, norm="in", proj_type="linear", down_scale=(2, 2, 2), kernel_size=(3, 3, 3)):
        super().__init__()
        if proj_type not in ("linear", "depthwise"):
            raise ValueError(proj_type)
        self.down_scale = _k3(down_scale)
        merged = 2 ** self.down_scale.count(2) * dim
        if any(s not in (1, 2) for s in self.down_scale):
            raise NotImplementedError("cbim_amd: PatchMerging scales other than 1/2 are not built")
        self.linear = proj_type == "linear"
        if self.linear:   # medformer_utils.py:153-154
            self.reduction = nn.Conv3d(merged, out_dim, kernel_size=1, bias=False)
        else:
            self.reduction = DepthwiseSeparableConv(merged, out_dim, kernel_size=kernel_size)
        self.norm_kind = norm if norm in ("bn", "ln") else "in"
        self.norm = make_norm(self.norm_kind, merged)   # `in`: InstanceNorm3d(merged), eps 1e-5, fused into the depthwise load

    def forward(self, f: Fn.FMap) -> Fn.FMap:
        m = Fn.SpaceToDepthFn.apply(f.t, tuple(self.down_scale))
        if self.norm_kind != "in":                      # `norm: bn | ln`: the normalised tensor written once, projections read it
            mn = norm_rows(self.norm, m)
            if self.linear:
                return Fn.FMap(Fn.NormConvFn.apply(mn, None, self.reduction.weight, 0, None, False, None, IN_EPS)[0], None)
            return self.reduction(mn, None, 0)
        ms = Fn.ensure_stats(Fn.FMap(m, None), _EPS_DEFAULT).stats
        if self.linear:
            y, so = Fn.NormConvFn.apply(m, ms, self.reduction.weight, 0, None, True, None, IN_EPS)
            return Fn.FMap(y, so)
        return self.reduction(m, ms, 0, want_stats=True)


class BasicLayer(nn.Module):
    """medformer_utils.py:177-200."""

    def __init__(self, feat_dim, map_dim, out_dim, num_blocks, heads=4, dim_head=64, expansion=4, attn_drop=0.,
                 proj_drop=0., map_size=(8, 8, 8), proj_type="depthwise", norm="in", act="gelu",
                 kernel_size=(3, 3, 3), no_map_out=False):
        super().__init__()
        blocks, d1 = [], feat_dim
        for i in range(num_blocks):
            blocks.append(BidirectionAttentionBlock(d1, map_dim, out_dim, heads, dim_head, expansion=expansion,
                                                    attn_drop=attn_drop, proj_drop=proj_drop, map_size=map_size,
                                                    proj_type=proj_type, norm=norm, act=act, kernel_size=kernel_size,
                                                    no_map_out=no_map_out if i == num_blocks - 1 else False))
            d1 = out_dim
        self.blocks = nn.ModuleList(blocks)

    def forward(self, f, semantic_map):
        for blk in self.blocks:
            f, semantic_map = blk(f, semantic_map)
        return f, semantic_map


class SemanticMapGeneration(nn.Module):
    """medformer_utils.py:203-228: the two 3^3 projections run as ONE Cout-concatenated convolution."""

    def __init__(self, feat_dim, map_dim, map_size):
        super().__init__()
        self.map_size, self.map_dim = list(map_size), map_dim
        self.map_code_num = map_size[0] * map_size[1] * map_size[2]
        self.base_proj = nn.Conv3d(feat_dim, map_dim, kernel_size=3, padding=1, bias=False)
        self.semantic_proj = nn.Conv3d(feat_dim, self.map_code_num, kernel_size=3, padding=1, bias=False)

    def forward(self, f: Fn.FMap):
        # the conv kernels move whole 16-byte channel chunks: a code count that is not a multiple of 8 (config/bcv:
        # map_size [3,3,3] = 27) is padded with zero-weight codes — their logits are 0, their pooled columns are cut
        # off below and receive no gradient, so the 27 real codes are untouched
        pad = (-self.map_code_num) % 8
        parts = [self.base_proj.weight, self.semantic_proj.weight]
        if pad:
            parts.append(self.semantic_proj.weight.new_zeros((pad,) + tuple(self.semantic_proj.weight.shape[1:])))
        w = torch.cat(parts, 0)
        fw, _ = Fn.NormConvFn.apply(f.t, None, w, 0, None, False, None, IN_EPS)
        mp = Fn.MapPoolFn.apply(fw, self.map_dim)                       # [B, map_dim, codes (+pad)]
        if pad:
            mp = mp[..., :self.map_code_num]
        return mp.reshape(mp.shape[0], self.map_dim, *self.map_size)


class SemanticMapFusion(nn.Module):
    """medformer_utils.py:231-261 — 3 x 64 tokens; torch ops (see trans_layers.py)."""

    def __init__(self, in_dim_list, dim, heads, depth=1, norm="in", attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.in_proj = nn.ModuleList([nn.Conv3d(c, dim, kernel_size=1, bias=False) for c in in_dim_list])
        self.fusion = TransformerBlock(dim, depth, heads, dim // heads, dim, attn_drop=attn_drop, proj_drop=proj_drop)
        self.out_proj = nn.ModuleList([nn.Conv3d(dim, c, kernel_size=1, bias=False) for c in in_dim_list])

    def forward(self, map_list):
        B, _, D, H, W = map_list[0].shape
        tok = torch.cat([pointwise(p, m).flatten(2).transpose(1, 2) for p, m in zip(self.in_proj, map_list)], dim=1)
        tok = self.fusion(tok)
        return [pointwise(p, t.transpose(1, 2).reshape(B, self.dim, D, H, W))
                for p, t in zip(self.out_proj, tok.chunk(len(map_list), dim=1))]


class inconv(nn.Module):
    """medformer_utils.py:264-277."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), block=BasicBlock, norm="in", act="gelu"):
        super().__init__()
        k = _k3(kernel_size)
        self.conv1 = nn.Conv3d(in_ch, out_ch, kernel_size=k, padding=[i // 2 for i in k], bias=False)
        self.conv2 = block(out_ch, out_ch, kernel_size=k, norm=norm, act=act)

    def forward(self, x, dtype) -> Fn.FMap:
        return self.conv2(Fn.FMap(Fn.StemFn.apply(x, self.conv1.weight, dtype), None))


class down_block(nn.Module):
    """medformer_utils.py:281-319."""

    def __init__(self, in_ch, out_ch, conv_num, trans_num, down_scale=(2, 2, 2), kernel_size=(3, 3, 3),
                 conv_block=BasicBlock, heads=4, dim_head=64, expansion=1, attn_drop=0., proj_drop=0.,
                 map_size=(8, 8, 8), proj_type="depthwise", norm="in", act="gelu", map_generate=False, map_dim=None):
        super().__init__()
        map_dim = out_ch if map_dim is None else map_dim
        self.map_generate = map_generate
        if map_generate:
            self.map_gen = SemanticMapGeneration(out_ch, map_dim, map_size)
        self.patch_merging = PatchMerging(in_ch, out_ch, norm=norm, proj_type=proj_type, down_scale=down_scale,
                                          kernel_size=kernel_size)
        self.conv_blocks = nn.Sequential(*[conv_block(out_ch, out_ch, norm=norm, act=act, kernel_size=kernel_size)
                                           for _ in range(conv_num)])
        self.trans_blocks = BasicLayer(out_ch, map_dim, out_ch, num_blocks=trans_num, heads=heads, dim_head=dim_head,
                                       norm=norm, act=act, expansion=expansion, attn_drop=attn_drop,
                                       proj_drop=proj_drop, map_size=map_size, proj_type=proj_type,
                                       kernel_size=kernel_size)

    def forward(self, f: Fn.FMap):
        f = self.patch_merging(f)
        for blk in self.conv_blocks:
            f = blk(f)
        semantic_map = self.map_gen(f) if self.map_generate else None
        return self.trans_blocks(f, semantic_map)


class up_block(nn.Module):
    """medformer_utils.py:321-372: cat([upsampled, skip]) (note the order), optional map shortcut."""

    def __init__(self, in_ch, out_ch, conv_num, trans_num, up_scale=(2, 2, 2), kernel_size=(3, 3, 3),
                 conv_block=BasicBlock, heads=4, dim_head=64, expansion=4, attn_drop=0., proj_drop=0.,
                 map_size=(4, 8, 8), proj_type="depthwise", norm="in", act="gelu", map_dim=None, map_shortcut=False,
                 no_map_out=False):
        super().__init__()
        self.map_shortcut = map_shortcut
        map_dim = out_ch if map_dim is None else map_dim
        self.map_reduction = nn.Conv3d(in_ch + out_ch, map_dim, kernel_size=1, bias=False) if map_shortcut else nn.Identity()
        self.trans_blocks = BasicLayer(in_ch + out_ch, map_dim, out_ch, num_blocks=trans_num, heads=heads,
                                       dim_head=dim_head, norm=norm, act=act, expansion=expansion, attn_drop=attn_drop,
                                       proj_drop=proj_drop, map_size=map_size, proj_type=proj_type,
                                       kernel_size=kernel_size, no_map_out=no_map_out)
        d1 = in_ch + out_ch if trans_num == 0 else out_ch
        convs = []
        for _ in range(conv_num):
            convs.append(conv_block(d1, out_ch, kernel_size=kernel_size, norm=norm, act=act))
            d1 = out_ch
        self.conv_blocks = nn.Sequential(*convs)

    def forward(self, low: Fn.FMap, skip: Fn.FMap, map1, map2=None):
        if self.map_shortcut and map2 is not None:
            semantic_map = pointwise(self.map_reduction, torch.cat([map1, map2], dim=1))
        else:
            semantic_map = map1
        convs = list(self.conv_blocks)
        if len(self.trans_blocks.blocks) == 0 and convs and Fn.fused_up_block(convs[0]):
            # conv-only decoder level (up3 / up4 of the shipped configs): the first block reads
            # a = relu(IN([up(low) | skip])) written in one pass; the concatenation is never stored
            first = convs[0]
            skip = Fn.ensure_stats(skip)
            want = len(convs) > 1
            out, so = Fn.UpBlockFirstFn.apply(low.t, skip.t, skip.stats, first.conv1.conv.weight, first.conv2.conv.weight,
                                              first.shortcut.conv.weight, first.conv1.act_code, want, False)
            f = Fn.FMap(out, so if want else None)
            convs = convs[1:]
        else:
            f = Fn.FMap(*Fn.UpCatFn.apply(low.t, skip.t, False, True))   # concat + InstanceNorm statistics in one pass
            f, semantic_map = self.trans_blocks(f, semantic_map)
        for blk in convs:
            f = blk(f)
        return f, semantic_map
