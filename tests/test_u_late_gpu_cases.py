"""-m gpu cases added after round 1's last GPU minute (correct on the host-side executor, first hardware run at the round-end
suite): they live in a file that sorts after every other test file so that `pytest -x` reaches the long-standing parity tests
(and the full-size configuration sweep) first."""
import pytest
import torch

from tests import op_checks as oc

pytestmark = pytest.mark.gpu

F32, BF16 = torch.float32, torch.bfloat16


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_mappool_odd_code_count(dev, dtype):
    oc.check_mappool(dev, dtype, N=1, C=40, M=27, dhw=(4, 5, 6))     # bcv map_size [3,3,3]: element-wise (one-wave) backward


def test_training_utils_surface(dev):
    from tests.optim_checks import check_training_utils_surface
    check_training_utils_surface(dev)


def test_medformer_bcv_structure_fp32_matches_reference_golden(dev):
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity("medformer_bcv_tiny", dev))


def test_medformer_bcv_structure_bf16_inside_envelope(dev):
    """0.39 per-tensor gradient-norm error on the executor (a 14-class net with InstanceNorm over 8 voxels at the deepest
    level; one small-norm tensor), logits 0.09 / 0.10."""
    from tests.medformer_checks import run_case
    r, g = run_case("medformer_bcv_tiny", dev, "bf16")
    print(r)
    assert r["logits_err"] < 0.4 and r["aux_err"] < 0.4, r          # (absolute backstop; the computed envelope follows)
    assert max(abs(a - b) for a, b in zip(r["ce"] + r["dice"], list(g["ce"]) + list(g["dice"]))) < 0.05, r
    assert r["grad_norm_err"] < 1.0, r
    from tests.test_gpu_parity import _medformer_envelope
    _medformer_envelope(dev, "medformer_bcv_tiny")


def test_resunet_bottleneck_matches_reference_golden(dev):
    from tests.model_checks import assert_fp32_parity, run_case
    print(assert_fp32_parity("resunet_bottleneck_b16", dev, max_flips=2, g_stem_tol=5e-2, grad_tol=0.15, cos_min=0.999))
    r, g = run_case("resunet_bottleneck_b16", dev, "bf16")
    print(r)
    # bf16 on untrained weights with three convs per block and InstanceNorm over 8 voxels at the deepest level: logits 0.73
    # (max-abs / max-abs) on the executor, yet CE 1.5315 vs 1.5256 and Dice 0.7636 vs 0.7615 — the losses are the criterion
    assert r["logits_err"] < 2.0 and abs(r["ce"] - float(g["ce"])) < 0.05 and abs(r["dice"] - float(g["dice"])) < 0.05, r



# ---- VNet ------------------------------------------------------------------------------------------------

def test_vnet_fp32_matches_reference_golden(dev):
    from tests.util import record_parity
    from tests.vnet_checks import assert_fp32
    r = assert_fp32(dev)
    print(r)
    record_parity("vnet_b8_fp32", {k: (float(v) if not isinstance(v, str) else v) for k, v in r.items()})


def test_vnet_bf16_inside_envelope(dev):
    """bf16 engine mode against the reference's fp32 golden (untrained weights, BatchNorm over 16 values at the deepest
    level): the losses are the criterion, the element-wise figures are recorded."""
    from tests.util import load_golden, record_parity
    from tests.vnet_checks import run
    r = run(dev, "bf16")
    print(r)
    record_parity("vnet_b8_bf16", {k: (float(v) if not isinstance(v, str) else v) for k, v in r.items()})
    g = load_golden("vnet_b8")
    assert r["ce_err"] < 0.05 and r["dice_err"] < 0.05, r
    assert r["logits_err"] < 0.5 and r["argmax_mismatch"] < 0.2 * g["logits"][:, 0].size, r      # (absolute backstop)
    # round 6: the computed envelope — oracle/vnet_ref.py in fp32 and under CPU autocast(bf16) with the golden's dropout masks, the
    # engine in bf16 with the same masks injected
    import contextlib
    from functools import partial
    from cbim_amd.model.dim3 import vnet as vmod
    from oracle import vnet_ref
    from tests.util import bf16_envelope_vs_oracle
    from tests.vnet_checks import SCALE, _golden_state_dict, _masks
    net, _ = _golden_state_dict(g)
    net = net.to(dev).train()

    @contextlib.contextmanager
    def injected_masks():
        masks, orig = _masks(g), vmod.dropout3d_mask
        vmod.dropout3d_mask = lambda n, c, p, training, device: masks.pop(0).to(device)
        try:
            yield
        finally:
            vmod.dropout3d_mask = orig

    def oracle(sd, x):
        return vnet_ref.vnet_forward(sd, x, SCALE, masks=_masks(g))

    env, bad = bf16_envelope_vs_oracle(dev, net, oracle, torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"]),
                                       tag="vnet_b8_bf16_envelope", engine_ctx=injected_masks)
    assert not bad, bad


def test_vnet_acdc_config_trains_in_bf16(dev):
    """config/acdc/vnet_3d.yaml's model (base 16) on its training crop 16 x 192 x 192, batch 2: three AdamW steps reduce the
    loss; every gradient finite"""
    import cbim_amd
    from cbim_amd.model.dim3 import VNet
    from cbim_amd.training.losses import DiceCELoss
    from cbim_amd.training.optim import FusedAdamW
    torch.manual_seed(5)
    net = VNet(1, 4, scale=[[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], baseChans=16).to(dev).train()
    w = torch.tensor([0.5, 1, 1, 1.0], device=dev)
    crit = DiceCELoss(w).to(dev)
    opt = FusedAdamW(net.parameters(), lr=1e-3, weight_decay=0.05)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 1, 16, 192, 192, generator=g).to(dev)
    lab = (torch.rand(2, 1, 2, 24, 24, generator=g) * 4).long().float()
    lab = torch.nn.functional.interpolate(lab, size=(16, 192, 192), mode="nearest").long().to(dev)
    cbim_amd.set_compute_dtype("bf16")
    try:
        losses = []
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            loss = crit(net(x), lab)
            loss.backward()
            assert all(torch.isfinite(p.grad).all() for p in net.parameters())
            opt.step()
            losses.append(float(loss))
    finally:
        cbim_amd.set_compute_dtype(None)
    print(losses)
    assert losses[-1] < losses[0], losses


def test_label_count_survives_graph_replay(dev):
    """a hipGraph-replayed training loop (the benchmarked mode) must keep counting out-of-range labels.  The device-side
    counter is allocated on the first eager loss call, the add is captured with the step, and check_labels() — or the validation
    entry points — raise what the reference's CrossEntropyLoss / scatter_ would have raised at the offending step."""
    import cbim_amd
    from cbim_amd.model.dim3 import UNet
    from cbim_amd.training.losses import DiceCELoss, check_labels
    from cbim_amd.training.optim import FusedAdamW
    cbim_amd.set_compute_dtype("bf16")
    try:
        torch.manual_seed(3)
        net = UNet(1, 8, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=4, block="BasicBlock", norm="in").to(dev)
        crit = DiceCELoss(torch.ones(4)).to(dev)
        opt = FusedAdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-5)
        x = torch.randn(1, 1, 32, 32, 32, device=dev)
        lab = torch.randint(0, 4, (1, 1, 32, 32, 32), device=dev)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(net(x), lab)
            loss.backward()
            opt.step()
            return loss

        check_labels()                               # re-arm
        for _ in range(6):                           # past the four on-the-spot checks: the counter path is live
            step()
        assert check_labels() == 0
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g, stream=side):
                step()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        assert check_labels() == 0                   # clean labels: nothing counted by capture + replay
        lab[0, 0, 0, 0, :3] = 7                      # three labels outside [0, 4) in the graph's static label buffer
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        with pytest.raises(IndexError, match="6 label"):
            check_labels()
        assert check_labels() == 0                   # reset by the raising call
    finally:
        cbim_amd.set_compute_dtype(None)


@pytest.mark.parametrize("name", ["resunet_bn_b8", "unet_single_bn_b8", "resunet_nopool_b8", "unet_single_nopool_bn",
                                  "resunet_bottleneck_nopool_b16", "resunet_ln_b8", "unet_single_ln_b8"])
def test_norm_bn_and_pool_false_branches_match_reference_golden(dev, name):
    """`norm: bn`, `norm: ln`, `pool=False` (round 5): fp32 parity with the real reference's UNet (training step + eval forward); the bf16 mode
    runs the same composed path — logits inside the usual bf16 distance, losses close."""
    from tests.bn_checks import assert_fp32, run_case
    from tests.util import record_parity
    r = assert_fp32(name, dev)
    print(r)
    record_parity("golden_" + name + "_fp32", r)
    rb, g = run_case(name, dev, "bf16")
    print(rb)
    record_parity("golden_" + name + "_bf16", rb)
    # (the Bottleneck pyramid — three convs per block, InstanceNorm over 8 voxels at the deepest level — is the fixture on which bf16
    #  logits of UNTRAINED weights scatter most, see test_resunet_bottleneck_matches_reference_golden: the losses are its criterion)
    lim = 2.0 if "bottleneck" in name else 0.25
    assert rb["logits_err"] < lim and abs(rb["ce"] - float(g["ce"])) < 0.05 and abs(rb["dice"] - float(g["dice"])) < 0.03, rb


def test_medformer_linear_projections_match_reference_golden(dev):
    """proj_type 'linear' (round 5): fp32 parity with the real reference, then the bf16 engine mode inside the MedFormer envelope."""
    from tests.medformer_checks import assert_fp32_parity, run_case
    from tests.util import record_parity
    r = assert_fp32_parity("medformer_linear_tiny", dev)
    print(r)
    record_parity("golden_medformer_linear_tiny_fp32", {k: v for k, v in r.items() if not isinstance(v, list)})
    rb, g = run_case("medformer_linear_tiny", dev, "bf16")
    print(rb)
    record_parity("golden_medformer_linear_tiny_bf16", {k: v for k, v in rb.items() if not isinstance(v, list)})
    assert rb["logits_err"] < 0.4 and rb["aux_err"] < 0.4, rb
    assert max(abs(a - b) for a, b in zip(rb["ce"] + rb["dice"], list(g["ce"]) + list(g["dice"]))) < 0.05, rb


@pytest.mark.parametrize("name", ["unetpp_bn_b8", "unetpp_ln_b8", "attunet_bn_b8", "attunet_ln_b8"])
def test_norm_branches_of_unetpp_and_attention_unet_match_reference_golden(dev, name):
    """UNet++ / AttentionUNet with `norm: bn | ln` (round 5; the reference constructors' own default is 'bn'): fp32 parity with one
    training step + the eval-mode forward of the real reference — the UNet++ BasicBlock fixture puts separate BatchNorms over the
    1-channel network input in front of conv1 and the shortcut (input gradient of the first convolution needed); AttentionUNet's gates
    keep nn.InstanceNorm3d.  bf16 mode: logits and losses (UNet++ fixtures; the base-8 AttentionUNet fixtures have a 4-channel gate
    projection, below the 8-channel chunk of the bf16 kernels — the shipped yaml's base 32 gives 16)."""
    from tests.norm_branch_checks import assert_norm_fp32, run_norm_case
    from tests.util import record_parity
    r = assert_norm_fp32(name, dev)
    print(r)
    record_parity("golden_" + name + "_fp32", r)
    if name.startswith("unetpp"):
        rb, g = run_norm_case(name, dev, "bf16")
        print(rb)
        record_parity("golden_" + name + "_bf16", rb)
        assert rb["logits_err"] < 0.25 and abs(rb["ce"] - float(g["ce"])) < 0.05 and abs(rb["dice"] - float(g["dice"])) < 0.03, rb
