"""Pins the oracle (oracle/) against fixtures produced by the REAL reference
(tests/golden/make_golden.py) and, where /root/reference is present, against the
reference executed live.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import loss_ref, unet_ref
from tests.util import CASES, golden_state_dict, load_golden, rel_err


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    in_ch, base, classes, scale, ks, block, shape, batch, seed = CASES[name]
    g = load_golden(name)
    sd = golden_state_dict(name)          # also checks seed->weights == reference constructor
    assert int(g["n_params"]) == sum(v.numel() for v in sd.values())
    assert int(g["n_tensors"]) == len(sd)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"])
    lab = torch.from_numpy(g["label"])
    w = torch.from_numpy(g["weight"])
    logits = unet_ref.unet_forward(sd, x, scale=scale, kernel_size=ks, block=block)
    assert rel_err(logits, g["logits"]) < 2e-5
    ce = loss_ref.cross_entropy(logits, lab.squeeze(1), w)
    dl = loss_ref.dice_loss(logits, lab)
    assert abs(float(ce) - float(g["ce"])) < 1e-5
    assert abs(float(dl) - float(g["dice"])) < 1e-5
    (ce + dl).backward()
    keys = [str(k) for k in g["keys"]]
    for i, k in enumerate(keys):
        gn = float(sd[k].grad.double().norm())
        assert abs(gn - g["grad_norms"][i]) <= 2e-3 * max(g["grad_norms"][i], 1e-6), (k, gn, g["grad_norms"][i])
    for k in ("inc.conv1.weight", "outc.weight", "outc.bias"):
        assert rel_err(sd[k].grad, g["g:" + k]) < 2e-3, k


def test_full_state_dict_fixture_roundtrip():
    g = load_golden("resunet_b2_32")
    sd = golden_state_dict("resunet_b2_32")
    for k, v in sd.items():
        np.testing.assert_array_equal(v.numpy(), g["p:" + k])


def test_pinned_facts():
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "facts.npz"))
    assert tuple(f["resunet_amos"]) == (40561008, 45, 0)
    assert tuple(f["unet_acdc"]) == (16266660, 20, 0)
    sd = unet_ref.make_unet_state_dict(1, 32, 16, [[3, 3, 3]] * 5, "BasicBlock", seed=0)
    assert sum(v.numel() for v in sd.values()) == 40561008 and len(sd) == 45
    torch.manual_seed(7)
    pred = torch.randn(2, 10, 8, 16, 16)
    target = torch.zeros(2, 1, 8, 16, 16).long()
    assert abs(float(loss_ref.dice_loss(pred, target)) - float(f["dice_zero_target"])) < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference/model/dim3"), reason="reference tree not mounted")
def test_oracle_vs_live_reference():
    from tests.golden.make_golden import import_reference
    UNet, DiceLoss = import_reference()
    torch.manual_seed(11)
    ks = [[3, 3, 3]] * 5
    sc = [[2, 2, 2]] * 4
    net = UNet(1, 4, scale=sc, kernel_size=ks, num_classes=5, block="BasicBlock", norm="in")
    x = torch.randn(1, 1, 32, 32, 32)
    lab = torch.randint(0, 5, (1, 1, 32, 32, 32))
    ref = net(x)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    out = unet_ref.unet_forward(sd, x, scale=sc, kernel_size=ks, block="BasicBlock")
    assert rel_err(out, ref.detach()) < 1e-5
    assert abs(float(DiceLoss()(ref, lab)) - float(loss_ref.dice_loss(out, lab))) < 1e-6
    # the two non-default reductions of the reference's DiceLoss (losses.py:48-56)
    assert abs(float(DiceLoss(size_average=False)(ref, lab)) - float(loss_ref.dice_loss(out, lab, size_average=False))) < 1e-5
    assert rel_err(loss_ref.dice_loss(out, lab, reduce=False), DiceLoss(reduce=False)(ref, lab).detach()) < 1e-5


def test_medformer_oracle_matches_reference_golden():
    """oracle/medformer_ref.py against outputs + every gradient of the REAL reference (fp32, CPU)."""
    from oracle.loss_ref import ce_dice_loss
    from oracle.medformer_ref import medformer_forward
    from tests.medformer_checks import TINY
    g = load_golden("medformer_tiny_32")
    sd = {k[2:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("p:")}
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    outs = medformer_forward(sd, x, map_size=TINY["map_size"], num_heads=TINY["num_heads"],
                             fusion_heads=TINY["fusion_heads"], fusion_depth=TINY["fusion_depth"],
                             kernel_size=TINY["kernel_size"], scale=TINY["scale"], act="relu", aux_loss=True)
    assert rel_err(outs[0], g["logits"]) < 1e-6 and rel_err(outs[1], g["aux_logits"]) < 1e-6
    loss = sum(0.5 * ce_dice_loss(o, lab, w) for o in outs)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    loss.backward()
    scale = float(np.max(g["grad_norms"]))
    for k, v in sd.items():
        r = torch.from_numpy(g["g:" + k])
        # bit-identical with the fixture's thread count; other thread counts reorder fp32 sums (the reference's
        # own fp32-vs-fp64 gradient noise on this case is ~1e-3)
        assert float((v.grad - r).abs().max()) <= 1e-3 * float(r.abs().max()) + 1e-7 * scale, k


def test_medformer_linear_oracle_matches_reference_golden():
    """proj_type 'linear' (1x1x1 projections, FusedMBConv feed-forward, linear PatchMerging): the oracle against the real reference's
    outputs + gradients.  The fixture stores no weights: they come from the seeded engine constructor, checksum-checked against the
    reference's state_dict (tests.medformer_checks.build)."""
    from oracle.loss_ref import ce_dice_loss
    from oracle.medformer_ref import medformer_forward
    from tests.medformer_checks import LIN_T, build
    net, g = build("medformer_linear_tiny", "cpu")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    outs = medformer_forward(sd, x, map_size=LIN_T["map_size"], num_heads=LIN_T["num_heads"],
                             fusion_heads=LIN_T["fusion_heads"], fusion_depth=LIN_T["fusion_depth"],
                             kernel_size=LIN_T["kernel_size"], scale=LIN_T["scale"], act="relu", aux_loss=True)
    st = int(g["stride"])
    assert rel_err(outs[0][..., ::st, ::st, ::st], g["logits"]) < 1e-5 and rel_err(outs[1][..., ::st, ::st, ::st], g["aux_logits"]) < 1e-5
    loss = sum(0.5 * ce_dice_loss(o, lab, w) for o in outs)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    scale = float(np.max(g["grad_norms"]))
    keys = [str(k) for k in g["keys"]]
    gn = np.array([float(sd[k].grad.double().norm()) for k in keys])
    assert float(np.max(np.abs(gn - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-5 * scale))) < 2e-2
    for k in keys:
        if "g:" + k in g.files:
            r = torch.from_numpy(g["g:" + k])
            assert float((sd[k].grad - r).abs().max()) <= 4e-2 * max(float(r.abs().max()), 1e-5 * scale), k


@pytest.mark.parametrize("name", ["medformer_bn_tiny", "medformer_ln_tiny"])
def test_medformer_norm_branches_oracle_matches_reference_golden(name):
    """`norm: bn` / `norm: ln` MedFormer (BatchNorm3d / channels-first LayerNorm in every ConvNormAct, norm1 / norm2 of the
    attention blocks, PatchMerging.norm): the oracle against the real reference's outputs, gradients and — BatchNorm — the running
    statistics its training-mode forward leaves.  Weights: the seeded engine constructor, checksum-checked (medformer_checks.build)."""
    from oracle.loss_ref import ce_dice_loss
    from oracle.medformer_ref import medformer_forward
    from tests.medformer_checks import MF_CASES, build
    net, g = build(name, "cpu")
    kw = MF_CASES[name][2]
    pnames = {k for k, _ in net.named_parameters()}
    sd = {k: (v.detach().clone().requires_grad_(True) if k in pnames else v.detach().clone()) for k, v in net.state_dict().items()}
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    outs = medformer_forward(sd, x, map_size=kw["map_size"], num_heads=kw["num_heads"], fusion_heads=kw["fusion_heads"],
                             fusion_depth=kw["fusion_depth"], kernel_size=kw["kernel_size"], scale=kw["scale"], act="relu", aux_loss=True)
    st = int(g["stride"])
    assert rel_err(outs[0][..., ::st, ::st, ::st], g["logits"]) < 1e-5 and rel_err(outs[1][..., ::st, ::st, ::st], g["aux_logits"]) < 1e-5
    loss = sum(0.5 * ce_dice_loss(o, lab, w) for o in outs)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    scale = float(np.max(g["grad_norms"]))
    keys = [str(k) for k in g["keys"]]
    gn = np.array([float(sd[k].grad.double().norm()) if k in pnames else 0.0 for k in keys])
    assert float(np.max(np.abs(gn - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-5 * scale))) < 2e-2
    for k in keys:
        if "g:" + k in g.files:
            r = torch.from_numpy(g["g:" + k])
            assert float((sd[k].grad - r).abs().max()) <= 4e-2 * max(float(r.abs().max()), 1e-5 * scale), k
        if "b:" + k in g.files:
            assert rel_err(sd[k].float(), g["b:" + k].astype(np.float32)) < 1e-5, k


def test_swin_oracle_matches_reference_golden():
    """oracle/swin_unetr_ref.py against the reference's swin_unetr.py executed on the monai stand-in
    (transformer part pinned; monai conv blocks parity-unpinned, see the oracle's header)."""
    from cbim_amd.model.dim3 import SwinUNETR          # only as the seeded weight initialiser (checksum-checked)
    from oracle.loss_ref import ce_dice_loss
    from oracle.swin_unetr_ref import swin_transformer, swin_unetr_forward
    from oracle.unet_ref import state_dict_checksum
    g = load_golden("swin_tiny")
    torch.manual_seed(int(g["seed"]))
    net = SwinUNETR((64, 32, 32), 4, 3, feature_size=24)
    pk = [k for k, _ in net.named_parameters()]
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert abs(state_dict_checksum({k: sd[k] for k in pk}) - float(g["sd_checksum"])) < 1e-6
    for k in pk:
        sd[k].requires_grad_(True)
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    hs = swin_transformer(sd, "swinViT.", x, (2, 2, 2, 0), (3, 6, 12, 24), (7, 7, 7))
    for i, h in enumerate(hs):
        assert rel_err(h, g[f"hidden{i}"]) < 1e-5, i
    logits = swin_unetr_forward(sd, x)
    assert rel_err(logits, g["logits"]) < 1e-5
    loss = ce_dice_loss(logits, lab, w)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    scale = float(np.max(g["grad_norms"]))
    for k in g.files:
        if k.startswith("g:"):
            r = torch.from_numpy(g[k])
            assert float((sd[k[2:]].grad - r).abs().max()) <= 1e-3 * float(r.abs().max()) + 1e-7 * scale, k


def test_inference_oracle_matches_reference_golden():
    """oracle/inference_ref.py (sliding window + Dice) against the real reference's outputs."""
    from functools import partial
    from oracle import inference_ref as IR
    g = load_golden("infer_resunet_b8")
    sd = unet_ref.make_unet_state_dict(1, 8, 3, [[3, 3, 3]] * 5, "BasicBlock", seed=int(g["seed"]))
    fwd = partial(unet_ref.unet_forward, sd, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, block="BasicBlock")
    with torch.no_grad():
        prob = IR.sliding_window(fwd, torch.from_numpy(g["x"]), [32, 32, 32], 3)
    assert rel_err(prob, g["prob"]) < 1e-5
    lp, lab = torch.from_numpy(g["label_pred"]), torch.from_numpy(g["label"])
    d, i, s = IR.dice(lp, lab, 3)
    assert np.array_equal(d.numpy(), g["dice"]) and np.array_equal(s.numpy(), g["summ"])
    d, i, s = IR.dice_split(lp, lab, 3, block_size=30000)
    assert np.array_equal(d.numpy(), g["dice_split"]) and np.array_equal(s.numpy(), g["summ_split"])


def test_unetpp_oracle_matches_reference_golden():
    from cbim_amd.model.dim3 import UNetPlusPlus      # seeded weight initialiser only (checksum-checked)
    from tests.unetpp_checks import KS, SCALE
    g = load_golden("unetpp_b8_acdc")
    torch.manual_seed(int(g["seed"]))
    sd = {k: v.detach().clone() for k, v in UNetPlusPlus(1, 8, scale=SCALE, kernel_size=KS, num_classes=4,
                                                        block="BasicBlock", norm="in").state_dict().items()}
    assert abs(unet_ref.state_dict_checksum(sd) - float(g["sd_checksum"])) < 1e-6
    with torch.no_grad():
        logits = unet_ref.unetpp_forward(sd, torch.from_numpy(g["x"]), scale=SCALE, kernel_size=KS, block="BasicBlock")
    assert rel_err(logits, g["logits"]) < 1e-5


@pytest.mark.parametrize("name", ["unetpp_bn_b8", "unetpp_ln_b8", "attunet_bn_b8", "attunet_ln_b8"])
def test_norm_branches_of_unetpp_and_attention_unet_oracle_matches_reference_golden(name):
    """UNet++ / AttentionUNet with `norm: bn | ln` (the reference constructors' own default is 'bn'): the oracle against one training
    step (logits, losses, gradient norms, full first-layer / gate / head gradients, running statistics) and the eval-mode logits of
    the real reference."""
    from tests.norm_branch_checks import NORM_CASES, build_norm_case, geometry
    net, g = build_norm_case(name)
    model, block, _ = NORM_CASES[name]
    fwd = unet_ref.unetpp_forward if model == "unetpp" else unet_ref.attention_unet_forward
    pk = [str(k) for k in g["param_keys"]]
    sd = {k: (v.detach().clone().requires_grad_(True) if k in pk else v.detach().clone()) for k, v in net.state_dict().items()}
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    lo = fwd(sd, x, block=block, training=True, **geometry(name))
    assert rel_err(lo.detach(), g["logits"]) < 2e-5
    ce, dl = loss_ref.cross_entropy(lo, lab.squeeze(1), w), loss_ref.dice_loss(lo, lab)
    assert abs(float(ce) - float(g["ce"])) < 1e-5 and abs(float(dl) - float(g["dice"])) < 1e-5
    (ce + dl).backward()
    scale = float(g["grad_norms"].max())
    for k, b in zip(pk, g["grad_norms"]):
        if b < 0:
            assert sd[k].grad is None, k
        else:
            assert abs(float(sd[k].grad.double().norm()) - b) / max(b, 1e-6 * scale) < 2e-3, k
    for k in g.files:
        if k.startswith("g:"):
            assert rel_err(sd[k[2:]].grad, g[k]) < 5e-3, k
        if k.startswith("r:"):
            assert rel_err(sd[k[2:]].double(), g[k].astype("float64")) < 1e-5, k
    with torch.no_grad():
        le = fwd(sd, x, block=block, training=False, **geometry(name))
    assert rel_err(le, g["logits_eval"]) < 2e-5


def test_attention_unet_oracle_matches_reference_golden():
    from tests.attunet_checks import KS, SCALE, build
    net, g = build()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        logits = unet_ref.attention_unet_forward(sd, torch.from_numpy(g["x"]), scale=SCALE, kernel_size=KS, block="BasicBlock")
    assert rel_err(logits, g["logits"]) < 1e-5


def test_vnet_oracle_matches_reference_golden():
    """oracle/vnet_ref.py against the real reference VNet in TRAINING mode with its recorded Dropout3d masks
    (tests/golden/make_golden_vnet.py): logits, loss, every gradient norm, the small gradient tensors in full."""
    from tests.vnet_checks import oracle_vs_golden
    print(oracle_vs_golden())


@pytest.mark.skipif(not os.path.isdir("/root/reference/model/dim3"), reason="reference tree not mounted")
def test_vnet_oracle_vs_live_reference():
    """another width / shape / seed, dropout off (eval: ContBatchNorm3d still normalises with batch statistics)"""
    import importlib
    from oracle import vnet_ref
    from tests.golden.make_golden import import_reference
    import_reference()
    VNet = importlib.import_module("model.dim3.vnet").VNet
    torch.manual_seed(23)
    sc = [[2, 2, 2]] * 4
    net = VNet(2, 3, scale=sc, baseChans=4).eval()
    x = torch.randn(2, 2, 16, 16, 32)
    ref = net(x)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    assert rel_err(vnet_ref.vnet_forward(sd, x, sc), ref.detach()) < 1e-5


BN_CASES = {
    # name: (in_ch, base_ch, classes, scale, kernel_size, block, seed, norm, pool)   (tests/golden/make_golden_bn.py)
    "resunet_bn_b8": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", 3031, "bn", True),
    "unet_single_bn_b8": (2, 8, 3, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "SingleConv", 3032, "bn", True),
    "resunet_nopool_b8": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", 3033, "in", False),
    "unet_single_nopool_bn": (1, 8, 3, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "SingleConv", 3034, "bn", False),
    "resunet_bottleneck_nopool_b16": (1, 16, 3, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "Bottleneck", 3035, "in", False),
    "resunet_ln_b8": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", 3036, "ln", True),
    "unet_single_ln_b8": (2, 8, 3, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "SingleConv", 3037, "ln", True),
}


def bn_state_dict(name):
    """the `norm: bn` state_dict of a golden case: reference-order keys from the seed + the fixture's perturbed affine parameters"""
    in_ch, base, classes, scale, ks, block, seed, norm, pool = BN_CASES[name]
    g = load_golden(name)
    sd = unet_ref.make_unet_state_dict(in_ch, base, classes, ks, block, seed=seed, norm=norm, pool=pool)
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    for k in g.files:
        if k.startswith("p:"):
            sd[k[2:]] = torch.from_numpy(g[k]).clone()
    chk = unet_ref.state_dict_checksum({k: v for k, v in sd.items() if v.is_floating_point()})
    assert abs(chk - float(g["sd_checksum"])) <= 1e-9 * max(1.0, abs(chk)), (chk, float(g["sd_checksum"]))
    return sd, g


@pytest.mark.parametrize("name", list(BN_CASES))
def test_oracle_batchnorm_branch_matches_reference_golden(name):
    """`norm: bn` (nn.BatchNorm3d in every ConvNormAct) and `pool=False` (strided first block per level): one training step and the
    eval-mode forward of the REAL reference."""
    in_ch, base, classes, scale, ks, block, seed, norm, pool = BN_CASES[name]
    sd, g = bn_state_dict(name)
    pk = [str(k) for k in g["param_keys"]]
    sdr = {k: (v.clone().requires_grad_(True) if k in pk else v.clone()) for k, v in sd.items()}
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    lo = unet_ref.unet_forward(sdr, x, scale=scale, kernel_size=ks, block=block, training=True, pool=pool)
    assert rel_err(lo.detach(), g["logits"]) < 2e-5
    ce, dl = loss_ref.cross_entropy(lo, lab.squeeze(1), w), loss_ref.dice_loss(lo, lab)
    assert abs(float(ce) - float(g["ce"])) < 1e-5 and abs(float(dl) - float(g["dice"])) < 1e-5
    (ce + dl).backward()
    gn = [float(sdr[k].grad.double().norm()) for k in pk]
    assert max(abs(a - b) / max(b, 1e-6) for a, b in zip(gn, g["grad_norms"])) < 2e-3
    for k in g.files:
        if k.startswith("g:"):
            assert rel_err(sdr[k[2:]].grad, g[k]) < 5e-3, k
        if k.startswith("r:"):
            assert rel_err(sdr[k[2:]].double(), g[k].astype("float64")) < 1e-5, k
    with torch.no_grad():
        le = unet_ref.unet_forward(sdr, x, scale=scale, kernel_size=ks, block=block, training=False, pool=pool)
    assert rel_err(le, g["logits_eval"]) < 2e-5
