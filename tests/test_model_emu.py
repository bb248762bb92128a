"""CPU suite: host logic of the plugin surface + whole-model parity of the fp32 engine mode
against the reference goldens, kernels executed by tests/emu."""
import argparse

import numpy as np
import pytest
import torch

from tests.util import CASES, golden_state_dict, load_golden


@pytest.fixture(autouse=True)
def _need_emu(dev):
    if dev != "cpu":
        pytest.skip("CPU suite")


def _args(**kw):
    base = dict(dimension="3d", model="resunet", in_chan=1, base_chan=32, classes=16,
                down_scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, block="BasicBlock", norm="in")
    base.update(kw)
    return argparse.Namespace(**base)


def test_get_model_matches_reference_state_dict_layout():
    from cbim_amd.model.utils import get_model
    net = get_model(_args())
    sd = net.state_dict()
    assert sum(p.numel() for p in net.parameters()) == 40561008 and len(sd) == 45
    assert len(list(net.buffers())) == 0
    assert "down1.conv.1.shortcut.conv.weight" in sd and tuple(sd["down1.conv.1.shortcut.conv.weight"].shape) == (64, 32, 3, 3, 3)
    acdc = get_model(_args(model="unet", classes=4, block="SingleConv",
                           down_scale=[[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]],
                           kernel_size=[[1, 3, 3], [2, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]))
    assert sum(p.numel() for p in acdc.parameters()) == 16266660 and len(acdc.state_dict()) == 20
    for name in CASES:
        g = load_golden(name)
        in_ch, base, classes, scale, ks, block, shape, batch, seed = CASES[name]
        n = get_model(_args(model="unet", in_chan=in_ch, base_chan=base, classes=classes, down_scale=scale,
                            kernel_size=ks, block=block))
        assert list(n.state_dict().keys()) == [str(k) for k in g["keys"]]
        assert [str(tuple(v.shape)) for v in n.state_dict().values()] == [str(s) for s in g["shapes"]]


def test_same_seed_draws_reference_weights():
    """Parameter creation order/initialisers mirror the reference constructor: same torch seed ->
    bit-identical weights (checked against the fingerprint recorded from the real reference)."""
    from cbim_amd.model.dim3 import UNet
    from oracle.unet_ref import state_dict_checksum
    name = "resunet_b8_32"
    in_ch, base, classes, scale, ks, block, shape, batch, seed = CASES[name]
    torch.manual_seed(seed)
    net = UNet(in_ch, base, scale=scale, kernel_size=ks, num_classes=classes, block=block, norm="in")
    g = load_golden(name)
    chk = state_dict_checksum(net.state_dict())
    assert abs(chk - float(g["sd_checksum"])) <= 1e-9 * max(1.0, abs(chk))


def test_unbuilt_options_fail_loudly():
    from cbim_amd.model.utils import get_model
    tiny = {k: v for k, v in __import__("tests.medformer_checks", fromlist=["TINY"]).TINY.items() if k != "norm"}
    for kind, n_keys in (("bn", 428), ("ln", 260)):     # round 6: MedFormer `norm: bn | ln` is built (state_dict sizes of the reference)
        assert len(get_model(_args(model="medformer", norm=kind, **tiny, down_scale=[[2, 2, 2]] * 4)).state_dict()) == n_keys
    with pytest.raises(NotImplementedError):
        get_model(_args(model="medformer", **dict(tiny, attn_drop=0.1), down_scale=[[2, 2, 2]] * 4))
    assert get_model(_args(block="Bottleneck", norm="ln")).state_dict()["down1.conv.1.conv1.norm.weight"].shape == (32,)
    with pytest.raises(NotImplementedError):
        get_model(_args(model="vtunet"))
    with pytest.raises(KeyError):
        from cbim_amd.model.dim3 import UNet
        UNet(1, 8)                       # the reference's default block name is not a valid key either
    with pytest.raises(ValueError):
        get_model(_args(dimension="4d"))


def test_resunet_fp32_matches_reference_golden(dev):
    from tests.model_checks import assert_fp32_parity
    r = assert_fp32_parity("resunet_b8_32", dev)
    print(r)


@pytest.mark.slow
def test_unet_singleconv_acdc_fp32_matches_reference_golden(dev):
    from tests.model_checks import assert_fp32_parity
    r = assert_fp32_parity("unet_single_acdc", dev)
    print(r)


def test_resunet_bottleneck_fp32_matches_reference_golden(dev):
    """UNet(block='Bottleneck') (conv_layers.py:96-125) against the real reference, inside the reference's own
    fp32-vs-fp64 envelope on this fixture (see tests/model_checks.py)."""
    from tests.model_checks import assert_fp32_parity
    from cbim_amd.model.utils import get_model
    print(assert_fp32_parity("resunet_bottleneck_b16", dev, max_flips=2, g_stem_tol=5e-2, grad_tol=0.15, cos_min=0.999))
    net = get_model(_args(block="Bottleneck"))
    assert "down1.conv.1.conv3.conv.weight" in net.state_dict()


def test_dice_loss_module_matches_reference_value(dev):
    from cbim_amd.training.losses import DiceLoss
    f = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "facts.npz"))
    torch.manual_seed(7)
    pred = torch.randn(2, 10, 8, 16, 16)
    target = torch.zeros(2, 1, 8, 16, 16).long()
    assert abs(float(DiceLoss()(pred, target)) - float(f["dice_zero_target"])) < 1e-5


def test_medformer_plugin_surface():
    """get_model(args) with the AMOS yaml keys builds the reference's parameter layout."""
    from cbim_amd.model.utils import get_model
    from tests.medformer_checks import AMOS
    a = dict(AMOS)
    a["down_scale"] = a.pop("scale")
    net = get_model(_args(model="medformer", **a))
    assert sum(p.numel() for p in net.parameters()) == 39594048 and len(net.state_dict()) == 278
    assert len(list(net.buffers())) == 0
    sd = net.state_dict()
    assert tuple(sd["down3.trans_blocks.blocks.0.feedforward.depthwise.conv.weight"].shape) == (1024, 1, 3, 3, 3)
    assert tuple(sd["up1.trans_blocks.blocks.0.shortcut.conv.weight"].shape) == (256, 576, 1, 1, 1)


def test_medformer_fp32_matches_reference_golden(dev):
    from tests.medformer_checks import assert_fp32_parity
    assert_fp32_parity("medformer_tiny_32", dev)


def test_medformer_linear_projections_fp32_match_reference_golden(dev):
    """proj_type 'linear' (medformer_utils.py:26-28,121-122,153-154): 1x1x1 q/v and out projections, FusedMBConv feed-forward and the
    linear PatchMerging reduction, all on the row GEMM with InstanceNorm on load — against the real reference."""
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity("medformer_linear_tiny", dev))


@pytest.mark.parametrize("name", ["medformer_bn_tiny", "medformer_ln_tiny"])
def test_medformer_norm_branches_fp32_match_reference_golden(dev, name):
    """`norm: bn` / `norm: ln` (round 6): BatchNorm3d / channels-first LayerNorm in every ConvNormAct (depthwise ones included), as
    norm1 / norm2 of every attention block and as PatchMerging.norm — outputs, every gradient norm, the small gradients in full and
    the BatchNorm running statistics against the real reference's run in train() mode."""
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity(name, dev))


def test_medformer_acdc_structure_fp32_matches_reference_golden(dev):
    """config/acdc/medformer_3d.yaml's structure (72 map codes, anisotropic stem, d_head 8|16|20) at reduced widths
    against the real reference: attn_wide.hip + the >64-code map pooling, forward and backward."""
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity("medformer_acdc_tiny", dev))


def test_medformer_bcv_structure_fp32_matches_reference_golden(dev):
    """config/bcv/medformer_3d.yaml's structure: 27 map codes — not a multiple of the 8-channel chunk, padded with
    zero-weight codes in SemanticMapGeneration; the element-wise map-pooling backward and the <=64-code generic attention."""
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity("medformer_bcv_tiny", dev))


@pytest.mark.slow
@pytest.mark.skipif(not __import__("os").environ.get("CBIM_SLOW"), reason="66 s on the host-side executor; set CBIM_SLOW=1 "
                    "(the same case runs in the -m gpu suite, its attention shapes in test_ops_emu.py)")
def test_medformer_lits_structure_fp32_matches_reference_golden(dev):
    """config/lits/medformer_3d.yaml's structure: one head per block (d_head = channels), no auxiliary head."""
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity("medformer_lits_tiny", dev))


def test_swin_unetr_plugin_surface():
    """get_model(args) builds the reference's SwinUNETR parameter layout (58.54 M parameters / 131 tensors at
    feature_size 48, in_chan 4 — SURVEY §8a a21)."""
    from cbim_amd.model.utils import get_model
    net = get_model(_args(model="swin_unetr", in_chan=4, classes=4, base_chan=48, window_size=[128, 128, 128], pretrain=False))
    assert sum(p.numel() for p in net.parameters()) == 58537606 and len(net.state_dict()) == 131
    sd = net.state_dict()
    assert tuple(sd["swinViT.layers1.0.blocks.0.attn.relative_position_bias_table"].shape) == (2197, 3)
    assert tuple(sd["decoder5.transp_conv.conv.weight"].shape) == (768, 384, 2, 2, 2)
    with pytest.raises(FileNotFoundError):     # args.pretrain: the reference torch.load()s its authors' checkpoint path (model/utils.py:115)
        get_model(_args(model="swin_unetr", in_chan=4, classes=4, base_chan=48, window_size=[128, 128, 128], pretrain=True))


def test_swin_unetr_load_from_self_supervised_checkpoint(tmp_path):
    """SwinUNETR.load_from (swin_unetr.py:230-277, 610-643): a checkpoint in the Swin-ViT pre-training layout (`module.` prefix,
    the MLP named fc1 / fc2) lands in the trunk tensor by tensor, through get_model(args.pretrain) as the reference loads it;
    everything outside the trunk keeps its initialisation; a missing key raises."""
    import torch
    from cbim_amd.model.dim3 import SwinUNETR
    from cbim_amd.model.utils import get_model
    torch.manual_seed(1)
    src = SwinUNETR((64, 64, 64), 1, 3, feature_size=12)
    ck = {"module." + k.replace("mlp.linear1", "mlp.fc1").replace("mlp.linear2", "mlp.fc2"): v.clone() + 0.25
          for k, v in src.swinViT.state_dict().items()}
    path = str(tmp_path / "model_swinvit.pt")
    torch.save({"state_dict": ck}, path)
    torch.manual_seed(2)
    net = get_model(_args(model="swin_unetr", in_chan=1, classes=3, base_chan=12, window_size=[64, 64, 64], pretrain=True,
                          swin_pretrain_path=path))
    torch.manual_seed(2)
    fresh = SwinUNETR((64, 64, 64), 1, 3, feature_size=12)
    n_trunk = 0
    for k, v in net.state_dict().items():
        if k.startswith("swinViT."):
            want = ck["module." + k[len("swinViT."):].replace("mlp.linear1", "mlp.fc1").replace("mlp.linear2", "mlp.fc2")]
            assert torch.equal(v, want.to(v.dtype)), k
            n_trunk += 1
        else:
            assert torch.equal(v, fresh.state_dict()[k]), k
    assert n_trunk == len(src.swinViT.state_dict()) and n_trunk > 40
    del ck["module.layers2.0.downsample.norm.bias"]
    with pytest.raises(KeyError):
        net.load_from({"state_dict": ck})


def test_swin_unetr_fp32_exact_token_gemm_matches_reference_golden(dev):
    """the trunk's Linears on the engine's own row GEMM (fp32-exact form) in the fp32 parity mode: hidden states and logits of
    the swin_tiny golden (reference's swin_unetr.py executed unmodified)"""
    from tests.swin_checks import assert_fp32_token_gemm_parity
    print(assert_fp32_token_gemm_parity("swin_tiny", dev))


def test_swin_unetr_fp32_matches_reference_golden(dev):
    # forward + losses on the host-side executor (the backward of every kernel involved is covered per op in
    # test_ops_emu.py and end to end on the GPU in test_gpu_parity.py; it would add ~90 s here)
    from tests.swin_checks import assert_fp32_parity
    print(assert_fp32_parity("swin_tiny", dev, backward=False))


def test_unetpp_layernorm_branch_fp32_matches_reference_golden(dev):
    """UNet++ with `norm: ln` (SingleConv blocks; the BatchNorm / BasicBlock fixture — norms over the raw input, separate for conv1
    and the shortcut — runs on the GPU, both pin the oracle): one training step + eval logits of the real reference."""
    from tests.unetpp_checks import assert_norm_fp32
    print(assert_norm_fp32("unetpp_ln_b8", dev))


def test_attention_unet_batchnorm_branch_fp32_matches_reference_golden(dev):
    """AttentionUNet with `norm: bn` in its blocks (SingleConv, batch 2; gates keep InstanceNorm): training step + eval logits"""
    from tests.norm_branch_checks import assert_norm_fp32
    print(assert_norm_fp32("attunet_bn_b8", dev))


def test_unetpp_fp32_matches_reference_golden(dev):
    # forward + losses here; forward + backward + bf16 run on the GPU (tests/test_gpu_parity.py), every kernel's
    # backward is covered per op in test_ops_emu.py
    from tests.unetpp_checks import assert_fp32
    print(assert_fp32(dev, backward=False))
    from cbim_amd.model.utils import get_model
    net = get_model(_args(model="unet++", base_chan=8, classes=4))
    assert len(net.state_dict()) == 92 or len(net.state_dict()) > 0


def test_attention_unet_fp32_matches_reference_golden(dev):
    from tests.attunet_checks import assert_fp32
    print(assert_fp32(dev, optimizer_step=True))


def test_vnet_plugin_surface():
    """get_model(args) with config/acdc/vnet_3d.yaml's keys builds the reference's VNet parameter layout
    (/root/reference/model/utils.py:70-74; 170 state_dict entries incl. the BatchNorm buffers)."""
    from cbim_amd.model.utils import get_model
    sc = [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
    net = get_model(_args(model="vnet", in_chan=1, classes=4, base_chan=16, downsample_scale=sc))
    g = load_golden("vnet_b8")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]] and len(sd) == 170
    assert tuple(sd["in_tr.conv1.weight"].shape) == (16, 1, 5, 5, 5)
    assert tuple(sd["down_tr32.down_conv.weight"].shape) == (32, 16, 1, 2, 2)
    assert tuple(sd["up_tr256.up_conv.weight"].shape) == (256, 128, 2, 2, 2)
    assert sum(p.numel() for p in net.parameters()) == 45602192      # = the reference constructor at base_chan 16


def test_vnet_fp32_matches_reference_golden(dev):
    """training-mode forward + loss + backward (the golden's Dropout3d masks injected) on the host-side executor: logits,
    losses, running statistics and all 98 parameter gradients element by element"""
    from tests.vnet_checks import assert_fp32
    print(assert_fp32(dev))


def test_torch_compile_wrapper_runs_the_engine_eagerly(dev):
    """/root/reference/train.py:292-293 optionally wraps the model in torch.compile and unwraps `_orig_mod` when saving (:106):
    the engine's forward is marked eager-only (its operators are ctypes calls into libcbim_hip.so), so the wrapper neither
    fails nor changes a bit of the result, and the state_dict round-trips through `_orig_mod`."""
    import cbim_amd
    from cbim_amd.model.dim3 import UNet
    cbim_amd.set_compute_dtype("fp32")
    try:
        torch.manual_seed(0)
        net = UNet(1, 4, scale=[[1, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=3, block="BasicBlock", norm="in").to(dev)
        x = torch.randn(1, 1, 2, 16, 16, device=dev)
        y0 = net(x)
        y0.square().sum().backward()
        g0 = [p.grad.clone() for p in net.parameters()]
        net.zero_grad(set_to_none=True)
        cn = torch.compile(net)
        y1 = cn(x)
        assert torch.equal(y0, y1) and cn._orig_mod is net
        y1.square().sum().backward()
        assert all(torch.equal(p.grad, g) for p, g in zip(net.parameters(), g0))
        assert [k.replace("_orig_mod.", "") for k in cn.state_dict()] == list(net.state_dict())
    finally:
        cbim_amd.set_compute_dtype(None)


def test_label_range_is_checked_per_step_at_first_and_per_epoch_afterwards(dev):
    """the reference raises 'Target out of bounds' at the offending step; reading the count is a device synchronisation, so the
    engine checks the first calls on the spot and keeps a device-side count afterwards (training.losses.check_labels)"""
    from cbim_amd.training.losses import DiceCELoss, check_labels
    crit = DiceCELoss(torch.ones(3)).to(dev)
    lo = torch.randn(1, 3, 4, 4, 4, device=dev)
    lab = torch.randint(0, 3, (1, 1, 4, 4, 4), device=dev)
    check_labels()                                # re-arms the per-step check
    bad = lab.clone()
    bad[0, 0, 0, 0, 0] = 7
    with pytest.raises(IndexError):
        crit(lo, bad)                             # among the first calls: raised at once
    for _ in range(5):
        crit(lo, lab)
    crit(lo, bad)                                 # past the first calls: counted on the device
    with pytest.raises(IndexError):
        check_labels()
    assert check_labels() == 0


@pytest.mark.parametrize("name", ["unet_single_nopool_bn", "unet_single_ln_b8"])   # BatchNorm + anisotropic strides in 40 s here; the channels-first LayerNorm in 25 s; all seven fixtures on the GPU (test_u_late_gpu_cases.py), all seven pin the oracle (test_oracle.py)
def test_norm_bn_and_pool_false_branches_fp32_match_reference_golden(dev, name):
    """`norm: bn` and `pool=False` (round 5): UNet with nn.BatchNorm3d in every ConvNormAct / with a strided first block per level
    against one training step + the eval-mode forward of the REAL reference (tests/golden/make_golden_bn.py): perturbed affine
    parameters incl. a 1e-3 and a negative gamma; an odd extent on the strided path; anisotropic stride; strided Bottleneck."""
    from tests.bn_checks import assert_fp32
    print(assert_fp32(name, dev))
