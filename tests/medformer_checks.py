"""Whole-model MedFormer parity checks against the goldens produced by the REAL reference
(tests/golden/make_golden_medformer.py) — shared by the CPU (host-side executor) and -m gpu suites."""
import numpy as np
import torch

import cbim_amd
from cbim_amd import functional as Fn
from cbim_amd.model.dim3 import MedFormer
from tests.util import load_golden, rel_err

TINY = dict(base_chan=8, map_size=[2, 2, 2], conv_block="BasicBlock", conv_num=[2, 1, 0, 0, 0, 1, 2, 2],
            trans_num=[0, 1, 1, 2, 1, 1, 0, 0], chan_num=[16, 16, 32, 40, 32, 16, 16, 8],
            num_heads=[1, 2, 4, 5, 4, 2, 1, 1], fusion_depth=2, fusion_dim=40, fusion_heads=5, expansion=4,
            attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu", kernel_size=[[3, 3, 3]] * 5,
            scale=[[2, 2, 2]] * 4, aux_loss=True)
AMOS = dict(base_chan=32, map_size=[4, 4, 4], conv_block="BasicBlock", conv_num=[2, 1, 0, 0, 0, 1, 2, 2],
            trans_num=[0, 1, 4, 6, 4, 1, 0, 0], chan_num=[64, 128, 256, 320, 256, 128, 64, 32],
            num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10, expansion=4,
            attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu", kernel_size=[[3, 3, 3]] * 5,
            scale=[[2, 2, 2]] * 4, aux_loss=True)
# the structure of config/acdc/medformer_3d.yaml (72 map codes, anisotropic stem, 4 heads: d_head 8|16|20) and of
# config/lits/medformer_3d.yaml (one head per block: d_head = channels, no auxiliary head) at reduced widths
ACDC_T = dict(base_chan=8, map_size=[2, 6, 6], conv_block="BasicBlock", conv_num=[2, 0, 0, 0, 0, 0, 2, 2],
              trans_num=[0, 2, 1, 1, 1, 1, 0, 0], chan_num=[16, 32, 64, 80, 64, 32, 16, 8],
              num_heads=[1, 4, 4, 4, 4, 4, 1, 1], fusion_depth=2, fusion_dim=32, fusion_heads=4, expansion=4,
              attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu",
              kernel_size=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
              scale=[[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]], aux_loss=True)
# config/bcv/medformer_3d.yaml's structure: map_size [3,3,3] = 27 codes (not a multiple of the 8-channel chunk: padded with
# zero-weight codes in SemanticMapGeneration), anisotropic stem, trans_num [0,2,4,6,4,2,0,0] cut to one block per level
BCV_T = dict(base_chan=8, map_size=[3, 3, 3], conv_block="BasicBlock", conv_num=[2, 0, 0, 0, 0, 0, 2, 2],
             trans_num=[0, 1, 1, 1, 1, 1, 0, 0], chan_num=[16, 32, 64, 80, 64, 32, 16, 8],
             num_heads=[1, 2, 4, 5, 4, 2, 1, 1], fusion_depth=2, fusion_dim=40, fusion_heads=5, expansion=4,
             attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu",
             kernel_size=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]],
             scale=[[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]], aux_loss=True)
LITS_T = dict(base_chan=8, map_size=[2, 2, 2], conv_block="BasicBlock", conv_num=[2, 0, 0, 0, 0, 0, 2, 2],
              trans_num=[0, 1, 1, 2, 1, 1, 0, 0], chan_num=[16, 32, 64, 80, 64, 32, 16, 8],
              num_heads=[1, 1, 1, 1, 1, 1, 1, 1], fusion_depth=2, fusion_dim=40, fusion_heads=5, expansion=4,
              attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu", kernel_size=[[3, 3, 3]] * 5,
              scale=[[2, 2, 2]] * 4, aux_loss=False)
LIN_T = dict(TINY, proj_type="linear")
# `norm: bn` / `norm: ln` (round 6; no shipped 3-D yaml): nn.BatchNorm3d / the channels-first LayerNorm in every ConvNormAct, as
# norm1 / norm2 of the attention blocks and as PatchMerging.norm
BN_T, LN_T = dict(TINY, norm="bn"), dict(TINY, norm="ln")
MF_CASES = {"medformer_bn_tiny": (1, 4, BN_T), "medformer_ln_tiny": (1, 4, LN_T),"medformer_tiny_32": (1, 4, TINY), "medformer_linear_tiny": (1, 4, LIN_T), "medformer_amos_64": (1, 16, AMOS),
            "medformer_acdc_tiny": (1, 4, ACDC_T), "medformer_lits_tiny": (1, 3, LITS_T),
            "medformer_bcv_tiny": (1, 14, BCV_T)}
AUX_WEIGHT = (0.5, 0.5)


def build(name, dev):
    """Same torch seed as the reference constructor -> bit-identical weights (fingerprint checked)."""
    from oracle.unet_ref import state_dict_checksum
    g = load_golden(name)
    in_ch, classes, kw = MF_CASES[name]
    torch.manual_seed(int(g["seed"]))
    net = MedFormer(in_ch, classes, **kw)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"]) and len(list(net.buffers())) == int(g["n_buffers"])
    chk = state_dict_checksum(sd)
    assert abs(chk - float(g["sd_checksum"])) <= 1e-9 * max(1.0, abs(chk)), (chk, float(g["sd_checksum"]))
    return net.to(dev), g


def run_case(name, dev, mode):
    cbim_amd.set_compute_dtype(mode)
    try:
        net, g = build(name, dev)
        x = torch.from_numpy(g["x"]).to(dev)
        lab = torch.from_numpy(g["label"]).to(dev)
        w = torch.from_numpy(g["weight"]).to(dev)
        outs = net(x)
        aux_on = isinstance(outs, (list, tuple))                 # medformer.py:98-101
        outs = list(outs) if aux_on else [outs]
        losses = [Fn.DiceCEFn.apply(o, lab, w) for o in outs]
        loss = sum(a * l[2] for a, l in zip(AUX_WEIGHT, losses)) if aux_on else losses[0][2]   # train.py:206-212
        loss.backward()
        st = int(g["stride"])
        params = dict(net.named_parameters())
        keys = [str(k) for k in g["keys"]]
        gn = np.array([float(params[k].grad.double().norm()) if k in params else 0.0 for k in keys])   # buffers: no gradient
        scale = float(np.max(g["grad_norms"]))
        res = {
            "logits_err": rel_err(outs[0].detach().cpu()[..., ::st, ::st, ::st], g["logits"]),
            "aux_err": rel_err(outs[-1].detach().cpu()[..., ::st, ::st, ::st], g["aux_logits"]),
            "ce": [float(l[0]) for l in losses], "dice": [float(l[1]) for l in losses], "loss": float(loss),
            # norms below 1e-5 of the largest are analytically-zero gradients (a bias in front of a normalisation:
            # 4e-8 of pure fp32 rounding noise in both implementations for map_fusion...fc2.bias of the ACDC case)
            "grad_norm_err": float(np.max(np.abs(gn - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-5 * scale))),
            "g_stem": rel_err(params["inc.conv1.weight"].grad.cpu(), g["g:inc.conv1.weight"]),
            "g_head": rel_err(params["outc.weight"].grad.cpu(), g["g:outc.weight"]),
            "g_aux": rel_err(params["aux_out.weight"].grad.cpu(), g["g:aux_out.weight"]) if aux_on else 0.0,
        }
        ref = torch.from_numpy(g["logits"])
        top2 = ref.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4          # ties below the fp32 noise floor are masked
        mine = outs[0].detach().cpu()[..., ::st, ::st, ::st].argmax(1)
        res["argmax_mismatch"] = int(((mine != ref.argmax(1)) & clear).sum())
        bufs = dict(net.named_buffers())
        if any(("b:" + k) in g.files for k in bufs):    # BatchNorm running statistics after the training-mode forward
            res["buffer_err"] = max(rel_err(bufs[k].detach().cpu().float(), g["b:" + k].astype(np.float32)) for k in bufs)
        stored = [k for k in keys if "g:" + k in g.files]
        if len(stored) > 3:                               # fixture with full gradients (all, or all small tensors)
            errs, meds = [], []
            for k in stored:
                r = torch.from_numpy(g["g:" + k]).double()
                d = (params[k].grad.detach().cpu().double() - r).abs()
                floor = max(float(r.abs().max()), 1e-5 * scale)
                errs.append(float(d.max()) / floor)
                meds.append(float(d.median()) / floor)
            res["grad_max_err"], res["grad_med_err"] = max(errs), max(meds)
        return res, g
    finally:
        cbim_amd.set_compute_dtype(None)


def assert_fp32_parity(name, dev):
    """north_star: within 1e-3 rel of the reference CPU path in fp32, argmax maps exact (ties masked)."""
    r, g = run_case(name, dev, "fp32")
    assert r["logits_err"] < 1e-3 and r["aux_err"] < 1e-3, r
    assert r["argmax_mismatch"] == 0, r
    assert max(abs(a - b) for a, b in zip(r["ce"] + r["dice"], list(g["ce"]) + list(g["dice"]))) < 1e-4, r
    assert abs(r["loss"] - float(g["loss"])) < 1e-4, r
    # gradients of a ReLU/InstanceNorm net in fp32: the reference itself is ~1e-3 (max-abs, per tensor) away from
    # its own fp64 evaluation, dominated by ReLU-mask flips at |x_hat| ~ 1e-6 that differ between implementations
    assert r["grad_norm_err"] < 2e-2 and r["g_stem"] < 2e-2 and r["g_head"] < 1e-3 and r["g_aux"] < 1e-3, r
    if "grad_max_err" in r:
        assert r["grad_max_err"] < 4e-2 and r["grad_med_err"] < 1e-2, r   # batch-1 fixture: InstanceNorm over 8 voxels at the deepest level
    if "buffer_err" in r:
        assert r["buffer_err"] < 1e-4, r
    return r
