"""Oracle: SwinUNETR forward (and, through torch autograd, backward).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Functional restatement over a ``state_dict`` with the reference's parameter names, stock ``torch`` ops on the
CPU, dtype-generic.  Two provenance classes:

* in-tree code of ``/root/reference/model/dim3/swin_unetr.py`` (window partition / shift / mask, WindowAttention
  with the relative-position table, SwinTransformerBlock, the v0.9 PatchMerging slice order, BasicLayer,
  proj_out) — restated from the file, line numbers cited, and PINNED: ``tests/golden/swin_tiny.npz`` is
  produced by executing that file unmodified (``tests/golden/make_golden_swin.py``);
* the ``monai==1.1.0`` blocks it imports (``UnetrBasicBlock``/``UnetResBlock``, ``UnetrUpBlock``, ``UnetOutBlock``,
  ``PatchEmbed``, ``MLPBlock``; swin_unetr.py:24-27) — monai is absent from this image, so these follow MONAI
  1.1.0's published behaviour as summarised in SURVEY.md §8c.  **Parity unpinned** for this part: the golden
  run uses the torch-only stand-in under ``tests/golden/monai_standin`` written from the same description.
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

EPS = 1e-5  # nn.InstanceNorm3d / nn.LayerNorm defaults


# ---------------------------------------------------------------------------------------------------
# in-tree transformer pieces
# ---------------------------------------------------------------------------------------------------

def effective_window(x_size, window_size, shift_size):
    """get_window_size (swin_unetr.py:358-381): a dimension not larger than the window uses itself, unshifted."""
    ws, ss = list(window_size), list(shift_size)
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            ss[i] = 0
    return tuple(ws), tuple(ss)


def partition(x, ws):
    """window_partition (:295-324): [b,d,h,w,c] -> [b*windows, tokens, c]."""
    b, d, h, w, c = x.shape
    x = x.reshape(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], c)


def unpartition(win, ws, dims):
    """window_reverse (:327-355)."""
    b, d, h, w = dims
    x = win.reshape(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(b, d, h, w, -1)


def region_mask(dims, ws, ss, dtype):
    """compute_mask (:737-773): 0 within a roll region, -100 across regions."""
    img = torch.zeros((1,) + tuple(dims) + (1,), dtype=dtype)
    cnt = 0
    for sd in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for sh in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for sw in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, sd, sh, sw, :] = cnt
                cnt += 1
    mw = partition(img, ws).squeeze(-1)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def relative_position_index(window_size):
    """WindowAttention.__init__ (:417-441): index into the (2w-1)^3 table for a full window."""
    coords = torch.stack(torch.meshgrid(*[torch.arange(w) for w in window_size], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window_size[0] - 1
    rel[:, :, 1] += window_size[1] - 1
    rel[:, :, 2] += window_size[2] - 1
    rel[:, :, 0] *= (2 * window_size[1] - 1) * (2 * window_size[2] - 1)
    rel[:, :, 1] *= 2 * window_size[2] - 1
    return rel.sum(-1)


def window_attention_core(qkv, table, rel_index, heads, mask):
    """WindowAttention.forward (:467-490) from the qkv projection to the head merge (proj excluded).
    qkv [bw, n, 3c]; table [T, heads]; mask [nw, n, n] or None."""
    bw, n, c3 = qkv.shape
    c = c3 // 3
    qkv = qkv.reshape(bw, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (c // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = table[rel_index[:n, :n].reshape(-1)].reshape(n, n, -1).permute(2, 0, 1)      # :473-476
    attn = attn + bias.unsqueeze(0)
    if mask is not None:                                                                  # :478-481
        nw = mask.shape[0]
        attn = (attn.reshape(bw // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).reshape(-1, heads, n, n)
    attn = F.softmax(attn, dim=-1)
    return (attn @ v).transpose(1, 2).reshape(bw, n, c)


def shifted_window_attention(x, sd, p, heads, window_size, shift_size, rel_index):
    """SwinTransformerBlock.forward_part1 (:554-606) without norm1: pad, roll, partition, attention, reverse,
    roll back, crop.  x is the norm1 output [b,d,h,w,c]."""
    b, d, h, w, c = x.shape
    ws, ss = effective_window((d, h, w), window_size, shift_size)
    pd, ph, pw = (ws[0] - d % ws[0]) % ws[0], (ws[1] - h % ws[1]) % ws[1], (ws[2] - w % ws[2]) % ws[2]
    x = F.pad(x, (0, 0, 0, pw, 0, ph, 0, pd))
    _, dp, hp, wp, _ = x.shape
    shifted = any(i > 0 for i in ss)
    mask = None
    if shifted:
        x = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        mask = region_mask((dp, hp, wp), ws, ss, x.dtype)
    win = partition(x, ws)
    qkv = F.linear(win, sd[p + "qkv.weight"], sd.get(p + "qkv.bias"))
    o = window_attention_core(qkv, sd[p + "relative_position_bias_table"], rel_index, heads, mask)
    o = F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])
    x = unpartition(o, ws, (b, dp, hp, wp))
    if shifted:
        x = torch.roll(x, shifts=ss, dims=(1, 2, 3))
    return x[:, :d, :h, :w, :]


def swin_block(sd, p, x, heads, window_size, shift_size, rel_index):
    """SwinTransformerBlock.forward (:645-656); MLPBlock = linear1 -> GELU -> linear2 (monai, names :640-643)."""
    c = x.shape[-1]
    h = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], EPS)
    x = x + shifted_window_attention(h, sd, p + "attn.", heads, window_size, shift_size, rel_index)
    h = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], EPS)
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.linear1.weight"], sd[p + "mlp.linear1.bias"])),
                 sd[p + "mlp.linear2.weight"], sd[p + "mlp.linear2.bias"])
    return x + h


def patch_merging_v09(sd, p, x):
    """PatchMerging.forward (:710-731): the v0.9 slice list — note x4..x6 repeat octants (1,0,1),(0,1,0),(0,0,1)."""
    b, d, h, w, c = x.shape
    if (d % 2) or (h % 2) or (w % 2):
        x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2, 0, d % 2))
    sel = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
    x = torch.cat([x[:, i::2, j::2, k::2, :] for i, j, k in sel], -1)
    x = F.layer_norm(x, (8 * c,), sd[p + "norm.weight"], sd[p + "norm.bias"], EPS)
    return F.linear(x, sd[p + "reduction.weight"])


def swin_transformer(sd, p, x, depths, num_heads, window_size, normalize=True):
    """SwinTransformer.forward (:985-997) on NCDHW input; returns the 5 hidden states as NCDHW."""
    rel_index = relative_position_index(window_size)
    shift = tuple(i // 2 for i in window_size)
    x = F.conv3d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=2)   # monai PatchEmbed
    x = x.permute(0, 2, 3, 4, 1)

    def out(t):   # proj_out (:970-983): layer_norm over channels without affine
        t = F.layer_norm(t, (t.shape[-1],)) if normalize else t
        return t.permute(0, 4, 1, 2, 3)

    outs = [out(x)]
    for li in range(4):
        q = f"{p}layers{li + 1}.0."
        for bi in range(depths[li]):                                                  # BasicLayer.forward (:843-873)
            x = swin_block(sd, f"{q}blocks.{bi}.", x, num_heads[li], window_size,
                           (0, 0, 0) if bi % 2 == 0 else shift, rel_index)
        x = patch_merging_v09(sd, q + "downsample.", x)
        outs.append(out(x))
    return outs


# ---------------------------------------------------------------------------------------------------
# monai 1.1.0 blocks (restated from the published behaviour; parity unpinned)
# ---------------------------------------------------------------------------------------------------

def _in(x):
    return F.instance_norm(x, eps=EPS)


def unet_res_block(sd, p, x):
    """monai UnetResBlock: conv3-IN-lrelu-conv3-IN, residual (1x1 conv + IN when channels change), add, lrelu."""
    out = F.leaky_relu(_in(F.conv3d(x, sd[p + "conv1.conv.weight"], None, 1, 1)), 0.01)
    out = _in(F.conv3d(out, sd[p + "conv2.conv.weight"], None, 1, 1))
    res = _in(F.conv3d(x, sd[p + "conv3.conv.weight"])) if (p + "conv3.conv.weight") in sd else x
    return F.leaky_relu(out + res, 0.01)


def unetr_up_block(sd, p, x, skip):
    """monai UnetrUpBlock: ConvTranspose3d(k=2,s=2,bias=False) -> cat([up, skip]) -> UnetResBlock."""
    up = F.conv_transpose3d(x, sd[p + "transp_conv.conv.weight"], None, stride=2)
    return unet_res_block(sd, p + "conv_block.", torch.cat((up, skip), dim=1))


def swin_unetr_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, *, depths=(2, 2, 2, 0), num_heads=(3, 6, 12, 24),
                       window_size=(7, 7, 7)) -> torch.Tensor:
    """SwinUNETR.forward (swin_unetr.py:279-292)."""
    hs = swin_transformer(sd, "swinViT.", x, depths, num_heads, window_size, True)
    enc0 = unet_res_block(sd, "encoder1.layer.", x)
    enc1 = unet_res_block(sd, "encoder2.layer.", hs[0])
    enc2 = unet_res_block(sd, "encoder3.layer.", hs[1])
    enc3 = unet_res_block(sd, "encoder4.layer.", hs[2])
    dec4 = unet_res_block(sd, "encoder10.layer.", hs[4])
    dec3 = unetr_up_block(sd, "decoder5.", dec4, hs[3])
    dec2 = unetr_up_block(sd, "decoder4.", dec3, enc3)
    dec1 = unetr_up_block(sd, "decoder3.", dec2, enc2)
    dec0 = unetr_up_block(sd, "decoder2.", dec1, enc1)
    out = unetr_up_block(sd, "decoder1.", dec0, enc0)
    return F.conv3d(out, sd["out.conv.conv.weight"], sd["out.conv.conv.bias"])
