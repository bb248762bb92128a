"""-m gpu: whole-model parity of the HIP engine on a real MI355X.

fp32 engine mode  : against the golden fixtures produced by the REAL reference (tests/golden) —
                    logits within 1e-3 rel, argmax maps exact, loss and gradients.
bf16 engine mode  : against the oracle's fp32 result inside the reference's own bf16-autocast
                    envelope (max rel 0.2, argmax agreement ~85 % on untrained weights).
full-size (128^3) : size-independent properties — determinism, finite loss/grads, loss decreases
                    under AdamW, hard-Dice of fp32 vs bf16 argmax maps.
"""
import pytest
import torch

from tests.model_checks import assert_fp32_parity, run_case
from tests.util import CASES, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for n in CASES if "bottleneck" not in n])   # Bottleneck: tests/test_u_late_gpu_cases.py
def test_fp32_mode_matches_reference_golden(dev, name):
    if CASES[name][1] % 4:
        pytest.skip("base_chan must be a multiple of 4 for whole 16-byte channel chunks")
    r = assert_fp32_parity(name, dev)
    print(name, r)
    from tests.util import record_parity
    record_parity("golden_" + name + "_fp32", dict(dtype="fp32", logits_rel=r["logits_err"], argmax_mismatch=r["argmax_mismatch"],
                                                  ce_abs=abs(r["ce"] - float(load_golden(name)["ce"])), grad_rel_worst=r["grad_rel_worst"],
                                                  grad_cos_min=r["grad_cos_min"], grad_norm_rel_worst=r["grad_norm_err"],
                                                  f64_ratio_worst=r["f64_ratio_worst"], f64_ratio_worst_tensor=str(r["f64_ratio_worst_tensor"]),
                                                  f64_err_engine=r["f64_err_engine"], f64_err_oracle32=r["f64_err_oracle32"],
                                                  f64_maxabs_ratio_worst=r["f64_maxabs_ratio_worst"]))


@pytest.mark.parametrize("name", ["resunet_b8_32", "resunet_b8_aniso", "unet_single_acdc"])
def test_bf16_mode_inside_reference_bf16_envelope(dev, name):
    r, g = run_case(name, dev, "bf16")
    print(name, r)
    from tests.util import record_parity
    env = r["bf16_envelope"]
    record_parity("golden_" + name + "_bf16", dict(dtype="bf16", logits_rel=r["logits_err"], argmax_mismatch=r["argmax_mismatch"],
                                                  n_vox=r["n_vox"], grad_rel_worst=r["grad_rel_worst"], grad_cos_min=r["grad_cos_min"],
                                                  **{"env_" + k: v for k, v in env.items()}))
    # the envelope is COMPUTED: the oracle under torch.autocast('cpu', bfloat16) on the same weights is the reference's own
    # reduced-precision run; the engine's logit error, argmax disagreements and every gradient's cosine deficit against the fp32
    # oracle must be no worse than 1.5 x that run's (tests.util.bf16_envelope) — untrained weights, so the numbers are large
    # on BOTH sides (cosines 0.5-0.95), which is exactly why a constant cannot be the bar
    assert not r["bf16_envelope_violations"], r["bf16_envelope_violations"]
    assert abs(r["ce"] - float(g["ce"])) < 0.05 and abs(r["dice"] - float(g["dice"])) < 0.02, r


def _net(dev, base=32, classes=16):
    from cbim_amd.model.dim3 import UNet
    torch.manual_seed(2023)
    return UNet(1, base, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=classes,
                block="BasicBlock", norm="in").to(dev)


def _data(dev, size, classes=16, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 1, size, size, size, generator=g).clamp_(-7.4, 2.2)
    coarse = torch.randint(0, classes, (1, 1, size // 8, size // 8, size // 8), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=(size,) * 3, mode="nearest").long()
    return x.to(dev), lab.to(dev)


def test_full_size_resunet_128_properties(dev):
    """BASELINE configs[1] shape: 1x1x128^3, base 32, 16 classes."""
    import cbim_amd
    from cbim_amd.training.losses import DiceCELoss
    from oracle.loss_ref import hard_dice
    x, lab = _data(dev, 128)
    w = torch.ones(16, device=dev)
    w[0] = 0.5
    crit = DiceCELoss(w).to(dev)
    outs = {}
    for mode in ("bf16", "fp32"):
        cbim_amd.set_compute_dtype(mode)
        net = _net(dev)
        l1 = net(x)
        loss = crit(l1, lab)
        loss.backward()
        g1 = [p.grad.clone() for p in net.parameters()]
        assert torch.isfinite(l1).all() and all(torch.isfinite(g).all() for g in g1)
        # determinism: same inputs -> bit-identical logits and gradients (fixed-order reductions)
        net.zero_grad(set_to_none=True)
        l2 = net(x)
        crit(l2, lab).backward()
        assert torch.equal(l1, l2)
        assert all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))
        outs[mode] = l1.detach()
    cbim_amd.set_compute_dtype(None)
    rel = float((outs["bf16"] - outs["fp32"]).abs().max() / outs["fp32"].abs().max())
    agree = float((outs["bf16"].argmax(1) == outs["fp32"].argmax(1)).float().mean())
    d32 = hard_dice(outs["fp32"].argmax(1).cpu(), lab.squeeze(1).cpu(), 16)
    d16 = hard_dice(outs["bf16"].argmax(1).cpu(), lab.squeeze(1).cpu(), 16)
    print("128^3 bf16 vs fp32 engine: max rel", rel, "argmax agreement", agree, "max |dDice|", float((d32 - d16).abs().max()))
    assert rel < 0.3 and agree > 0.75          # the reference's own bf16-autocast envelope


def test_training_reduces_loss_and_matches_oracle_at_64(dev):
    """A few AdamW steps at 64^3 (base 16): loss goes down in both engine modes; the fp32 engine's
    first-step loss and logits match the CPU oracle on the same weights and inputs."""
    import cbim_amd
    from cbim_amd.training.losses import DiceCELoss
    from oracle import loss_ref, unet_ref
    x, lab = _data(dev, 64, classes=8, seed=3)
    w = torch.ones(8, device=dev)
    w[0] = 0.5
    ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
    sd = unet_ref.make_unet_state_dict(1, 16, 8, ks, "BasicBlock", seed=77)
    lo = unet_ref.unet_forward(sd, x.cpu(), scale=sc, kernel_size=ks, block="BasicBlock")
    l_or = float(loss_ref.ce_dice_loss(lo, lab.cpu(), w.cpu()))
    for mode in ("fp32", "bf16"):
        cbim_amd.set_compute_dtype(mode)
        from cbim_amd.model.dim3 import UNet
        net = UNet(1, 16, scale=sc, kernel_size=ks, num_classes=8, block="BasicBlock", norm="in").to(dev)
        net.load_state_dict(sd)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, eps=1e-5)
        crit = DiceCELoss(w).to(dev)
        losses = []
        for i in range(6):
            opt.zero_grad(set_to_none=True)
            logits = net(x)
            if i == 0 and mode == "fp32":
                err = float((logits.detach().cpu() - lo).abs().max() / lo.abs().max())
                # argmax maps must be identical wherever the oracle's top-2 logit gap is not an fp32
                # tie (< 1e-4: the reference's own fp32-vs-fp64 run flips such voxels)
                top2 = lo.topk(2, dim=1).values
                decided = (top2[:, 0] - top2[:, 1]) > 1e-4
                diff = logits.argmax(1).cpu() != lo.argmax(1)
                mism = int((diff & decided).sum())
                print("64^3 fp32 vs oracle: logits rel err", err, "argmax mismatches", int(diff.sum()),
                      "of which outside fp32 ties", mism, "tie voxels", int((~decided).sum()))
                assert err < 1e-3 and mism == 0 and int(diff.sum()) <= 1e-4 * diff.numel()
                dice_o = loss_ref.hard_dice(lo.argmax(1), lab.squeeze(1).cpu(), 8)
                dice_h = loss_ref.hard_dice(logits.argmax(1).cpu(), lab.squeeze(1).cpu(), 8)
                assert float((dice_o - dice_h).abs().max()) < 0.002
            loss = crit(logits, lab)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        print(mode, losses)
        assert abs(losses[0] - l_or) < (1e-4 if mode == "fp32" else 0.05)
        assert losses[-1] < losses[0]
    cbim_amd.set_compute_dtype(None)


def test_smoke_entry(dev):
    import __graft_entry__ as ge
    ge.smoke()


# ---- MedFormer -------------------------------------------------------------

@pytest.mark.parametrize("name", ["medformer_tiny_32", "medformer_amos_64", "medformer_acdc_tiny", "medformer_lits_tiny"])
def test_medformer_fp32_matches_reference_golden(dev, name):
    from tests.medformer_checks import assert_fp32_parity as mf_parity
    print(name, mf_parity(name, dev))


def test_medformer_bf16_inside_envelope(dev):
    """bf16 engine mode of the shipped AMOS config at 64^3 against the reference's fp32 golden: same
    envelope as the UNet family."""
    from tests.medformer_checks import run_case as mf_run
    r, g = mf_run("medformer_amos_64", dev, "bf16")
    print(r)
    # the reference itself under torch.autocast(bfloat16) on this case (same weights, CPU): max rel 0.366 (logits) /
    # 0.235 (aux), 23 % argmax flips — measured with tests/golden/make_golden_medformer.py's model
    # absolute backstop (the relative envelope alone passes a regression that also degrades the autocast oracle)
    assert r["logits_err"] < 0.4 and r["aux_err"] < 0.4, r
    assert max(abs(a - b) for a, b in zip(r["ce"] + r["dice"], list(g["ce"]) + list(g["dice"]))) < 0.05, r
    assert r["grad_norm_err"] < 0.5, r
    # round 6: the bar itself is COMPUTED — the oracle under CPU autocast(bf16) on the same weights is the reference's own
    # reduced-precision run; logits, argmax flips and every gradient tensor's cosine deficit within 1.5 x of it
    _medformer_envelope(dev, "medformer_amos_64")


def _medformer_envelope(dev, name):
    from functools import partial
    from oracle.medformer_ref import medformer_forward
    from tests.medformer_checks import AUX_WEIGHT, MF_CASES, build
    from tests.util import bf16_envelope_vs_oracle
    net, g = build(name, dev)
    m = MF_CASES[name][2]
    fwd = partial(medformer_forward, map_size=m["map_size"], num_heads=m["num_heads"], fusion_heads=m["fusion_heads"],
                  fusion_depth=m["fusion_depth"], kernel_size=m["kernel_size"], scale=m["scale"], act=m["act"], aux_loss=m["aux_loss"])
    env, bad = bf16_envelope_vs_oracle(dev, net, fwd, torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"]),
                                       tag=name + "_bf16_envelope", loss_weights=AUX_WEIGHT)
    assert not bad, bad
    return env


@pytest.mark.parametrize("name", ["medformer_acdc_tiny", "medformer_lits_tiny"])
def test_medformer_wide_heads_bf16_inside_envelope(dev, name):
    """bf16 engine mode of the ACDC- (72 map codes, d_head 8|16|20) and LiTS-structured (one head per block, d_head up
    to 80, no auxiliary head) configurations: attn_wide.hip and the >64-code map pooling in bf16.  The same cases on
    the host-side executor: logits 0.09 / 0.15, losses within 1e-3, gradient norms 0.17 / 0.13."""
    from tests.medformer_checks import run_case as mf_run
    r, g = mf_run(name, dev, "bf16")
    print(r)
    assert r["logits_err"] < 0.4 and r["aux_err"] < 0.4, r          # (absolute backstop; the computed envelope follows)
    assert max(abs(a - b) for a, b in zip(r["ce"] + r["dice"], list(g["ce"]) + list(g["dice"]))) < 0.05, r
    assert r["grad_norm_err"] < 0.5, r
    _medformer_envelope(dev, name)


@pytest.mark.parametrize("name", ["medformer_bn_tiny", "medformer_ln_tiny"])
def test_medformer_norm_branches_match_reference_golden(dev, name):
    """`norm: bn` / `norm: ln` MedFormer (round 6; medformer_utils.py:112-113,119,122-124,158 with model/dim3/utils.py:15-21): fp32
    engine mode against the real reference's train()-mode run (logits, losses, every gradient norm, the small gradients in full,
    BatchNorm running statistics)."""
    from tests.medformer_checks import assert_fp32_parity as mf_parity
    print(name, mf_parity(name, dev))


@pytest.mark.parametrize("name", ["medformer_bn_tiny", "medformer_ln_tiny", "medformer_tiny_32"])
def test_medformer_norm_branches_bf16_inside_envelope_over_eight_inputs(dev, name):
    """bf16 engine mode of the `bn` / `ln` branches (and the `in` model of the same widths beside them) inside 1.5 x the oracle's own
    autocast(bf16) deviation — pooled over 8 inputs (the golden's + 7 seeded ones): at these widths the deepest level holds 2^3
    voxels and the single-input statistic is a coin flip on a few tensors for ALL three models (tests.util.bf16_envelope_samples
    has the measurements)."""
    from functools import partial
    from oracle.medformer_ref import medformer_forward
    from tests.golden.make_golden import make_labels
    from tests.medformer_checks import AUX_WEIGHT, MF_CASES, build
    from tests.util import bf16_envelope_samples
    _, g = build(name, "cpu")
    classes, m = MF_CASES[name][1], MF_CASES[name][2]
    fwd = partial(medformer_forward, map_size=m["map_size"], num_heads=m["num_heads"], fusion_heads=m["fusion_heads"],
                  fusion_depth=m["fusion_depth"], kernel_size=m["kernel_size"], scale=m["scale"], act=m["act"], aux_loss=m["aux_loss"])
    samples = [(torch.from_numpy(g["x"]), torch.from_numpy(g["label"]))]
    for seed in range(11, 18):
        gen = torch.Generator().manual_seed(seed)
        samples.append((torch.randn((1, 1, 32, 32, 32), generator=gen).clamp_(-7.4, 2.2), make_labels(classes, (32, 32, 32), 1, gen)))
    env, bad = bf16_envelope_samples(dev, lambda: build(name, dev)[0], fwd, samples, torch.from_numpy(g["weight"]),
                                     tag=name + "_bf16_envelope_8_inputs", loss_weights=AUX_WEIGHT)
    assert not bad, bad


@pytest.mark.parametrize("cfg,size,n_gemm", [("lits/medformer_3d.yaml", (64, 64, 64), 12), ("acdc/medformer_3d.yaml", (8, 96, 96), None)])
def test_medformer_wide_configs_on_the_gemm_attention_path_bf16_inside_envelope(dev, cfg, size, n_gemm):
    """config/lits/medformer_3d.yaml (num_heads all 1: d_head = 128 / 256 / 320, 64 map codes; 64^3 input) and config/acdc/medformer_3d.yaml
    (4 heads, 72 codes, anisotropic stem; 8x96x96 input) at their shipped widths: the attention cores of the levels with >= 512 voxels
    run as matrix products (round 6: ops.bidir_attn_gemm_*, csrc/attn_gemm_kernels.hip; the coarsest level stays on attn_wide.hip).  bf16 engine inside 1.5 x the oracle's own autocast(bf16) deviation; the same step on the
    vector-ALU kernels it replaces is recorded beside it."""
    import json
    import os
    from functools import partial
    import cbim_amd
    from cbim_amd import functional as Fn
    from cbim_amd import ops
    from cbim_amd.model.dim3 import MedFormer
    from oracle.medformer_ref import medformer_forward
    from tests.util import bf16_envelope_vs_oracle
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shipped_configs.json")) as f:
        a = json.load(f)[cfg]["args"]
    kw = {k: a[k] for k in ("base_chan", "map_size", "conv_block", "conv_num", "trans_num", "num_heads", "fusion_depth", "fusion_dim",
                            "fusion_heads", "expansion", "proj_type", "norm", "act", "kernel_size", "aux_loss")}
    kw.update(chan_num=a.get("chan_num", [64, 128, 256, 320, 256, 128, 64, 32]), scale=a["down_scale"], attn_drop=0., proj_drop=0.)
    fwd = partial(medformer_forward, map_size=kw["map_size"], num_heads=kw["num_heads"], fusion_heads=kw["fusion_heads"],
                  fusion_depth=kw["fusion_depth"], kernel_size=kw["kernel_size"], scale=kw["scale"], act=kw["act"], aux_loss=kw["aux_loss"])
    classes = int(a["classes"])
    g = torch.Generator().manual_seed(41)
    coarse = torch.randint(0, classes, (1, 1) + tuple(max(1, e // 8) for e in size), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=tuple(size), mode="nearest").long()
    x = torch.randn((1, 1) + tuple(size), generator=g).clamp_(-7.4, 2.2)
    w = torch.ones(classes)
    w[0] = 0.5
    torch.manual_seed(2024)
    net = MedFormer(1, classes, **kw).to(dev)
    taken = []
    orig = ops.bidir_attn_gemm_fwd
    ops.bidir_attn_gemm_fwd = lambda *args: (taken.append(int(args[0].shape[1])), orig(*args))[1]
    try:
        env, bad = bf16_envelope_vs_oracle(dev, net, fwd, x, lab, w, tag="medformer_" + cfg.split("/")[0] + "_gemm_attention_bf16_envelope",
                                           loss_weights=a.get("aux_weight") if kw["aux_loss"] else None)
    finally:
        ops.bidir_attn_gemm_fwd = orig
    print("attention cores on the GEMM path:", len(taken), "calls")
    assert len(taken) > 0 and (n_gemm is None or len(taken) == n_gemm), taken      # lits: 2 + 4 (down) + 4 + 2 (up) blocks at 16^3 / 8^3
    assert not bad, bad
    # the same bf16 step on attn_wide.hip (fp32 accumulation of the same bf16 operands)
    def step():
        for p in net.parameters():
            p.grad = None
        cbim_amd.set_compute_dtype("bf16")
        try:
            out = net(x.to(dev))
            outs = list(out) if isinstance(out, (list, tuple)) else [out]
            sum(Fn.DiceCEFn.apply(o, lab.to(dev), w.to(dev))[2] for o in outs).backward()
            out = outs[0]
        finally:
            cbim_amd.set_compute_dtype(None)
        return out.detach().float().cpu(), {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}
    lg1, g1 = step()
    ops.AWG = False
    try:
        lg0, g0 = step()
    finally:
        ops.AWG = True
    e = float((lg1 - lg0).abs().max() / lg0.abs().max())
    cos = min(float(torch.dot(g1[k].flatten().double(), g0[k].flatten().double()) / (g1[k].norm().double() * g0[k].norm().double()).clamp_min(1e-300))
              for k in g0 if g0[k].numel() >= 64 and float(g0[k].abs().max()) > 0)
    print(f"gemm attention vs attn_wide, same bf16 step: logits {e:.3e}, lowest gradient cosine {cos:.4f}")
    from tests.util import record_parity
    # (a record, not a bar: on this untrained 30-block net two bf16 evaluations differ from each other about as much as each
    #  differs from the fp32 oracle — measured 0.17 of the logit range between them against 0.30 / 0.34 to the oracle for the
    #  engine / the oracle's own autocast run; the kernels are compared on identical operands in tests/op_checks.check_attn_gemm)
    record_parity("medformer_" + cfg.split("/")[0] + "_gemm_vs_wide", {"logits_rel": e, "grad_cos_min": cos})


# ---- SwinUNETR -------------------------------------------------------------

@pytest.mark.parametrize("name", ["swin_tiny", "swin_brats_64", "swin_c1_tiny"])
def test_swin_unetr_fp32_matches_reference_golden(dev, name):
    from tests.swin_checks import assert_fp32_parity as sw_parity
    print(name, sw_parity(name, dev))


@pytest.mark.parametrize("name", ["swin_tiny", "swin_brats_64"])
def test_swin_unetr_fp32_exact_token_gemm_matches_reference_golden(dev, name):
    """the in-tree token GEMM (k_conv_pw token mode, fp32-exact form) instead of F.linear in the fp32 parity mode"""
    from tests.swin_checks import assert_fp32_token_gemm_parity
    r = assert_fp32_token_gemm_parity(name, dev)
    print(name, r)
    from tests.util import record_parity
    record_parity(name + "_fp32_token_gemm", r)


def test_swin_unetr_bf16_inside_envelope(dev):
    from tests.swin_checks import run_case as sw_run
    r, g = sw_run("swin_brats_64", dev, "bf16")
    print(r)
    assert r["logits_err"] < 0.25 and r["argmax_mismatch"] < 0.2 * r["n_vox"], r     # (absolute backstop)
    assert abs(r["ce"] - float(g["ce"])) < 0.05 and abs(r["dice"] - float(g["dice"])) < 0.02, r
    assert r["grad_norm_err"] < 0.5, r
    # round 6: the computed envelope (oracle under CPU autocast(bf16) = the reference's own reduced-precision run); the 64^3
    # feature-48 case runs the 48-channel convolution kernels (k_conv3_rw48, k_wgrad_r32 on 16-multiples) inside the model
    from oracle.swin_unetr_ref import swin_unetr_forward
    from tests.swin_checks import build as sw_build
    from tests.util import bf16_envelope_vs_oracle
    net, g = sw_build("swin_brats_64", dev)
    env, bad = bf16_envelope_vs_oracle(dev, net, swin_unetr_forward, torch.from_numpy(g["x"]), torch.from_numpy(g["label"]),
                                       torch.from_numpy(g["weight"]), tag="swin_brats_64_bf16_envelope")
    assert not bad, bad


# ---- sliding-window inference + evaluation Dice -------------------------------

def test_sliding_window_inference_and_dice(dev):
    from tests import infer_checks as ic
    ic.check_dice_exact(dev)
    ic.check_sliding_window(dev)


# ---- ragged / non-cubic volumes and batch > 1 against the oracle evaluated on the host ------------------

F64_MAX_VOXELS = 64 ** 3 * 4     # the float64 oracle is evaluated up to this many input elements (a 4-modality 64^3 volume)


def _oracle_vs_engine(dev, net, oracle_forward, x, lab, w, aux=False, tag=None, grad_tol=1e-1, f64=None):
    """fp32 engine mode vs the oracle (stock torch on the CPU) on the same weights: logits, loss, and EVERY parameter gradient.
    Both sides are fp32 evaluations of a deep network, so neither is the truth: up to 64^3 (f64, default by size) the oracle is
    ALSO evaluated in float64 and every gradient tensor of the engine must be at most 4x as far (L2) from that truth as the
    stock-torch fp32 evaluation is (tests.util.f64_bar: measured 0.1x - 3.0x) — this replaces the blanket element-wise 1e-1 of round 4.  Above 64^3
    (the 128^3 benchmarked shapes, where a float64 CPU evaluation takes minutes) the engine-vs-fp32-oracle numbers are
    recorded, the cosine (>= 0.999: what a permuted / transposed / sign-flipped gradient cannot pass) and the norms (2 %) are
    asserted, and the element-wise distance is bounded by `grad_tol` as an outlier guard only — the same architectures carry
    the float64 bar at 64^3 (test_*_64_fp32_engine_inside_the_f64_bar)."""
    if f64 is None:
        f64 = x.numel() <= F64_MAX_VOXELS
    import cbim_amd
    from cbim_amd import functional as Fn
    from oracle.loss_ref import ce_dice_loss
    from tests.util import grad_compare, record_parity
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    outs = oracle_forward(sd, x)
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    loss_ref = sum(ce_dice_loss(o, lab, w) for o in outs) / len(outs)
    loss_ref.backward()
    cbim_amd.set_compute_dtype("fp32")
    try:
        res = net(x.to(dev))
        res = res if isinstance(res, (list, tuple)) else [res]
        loss = sum(Fn.DiceCEFn.apply(o, lab.to(dev), w.to(dev))[2] for o in res) / len(res)
        loss.backward()
    finally:
        cbim_amd.set_compute_dtype(None)
    e_logits = 0.0
    for o, r in zip(res, outs):
        e = float((o.detach().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        e_logits = max(e_logits, e)
        assert e < 1e-3, e
    assert abs(float(loss) - float(loss_ref)) < 1e-4
    scale = max(float(v.grad.norm()) for v in sd.values() if v.grad is not None)
    e_norm = 0.0
    for k, p in net.named_parameters():
        a, b = float(p.grad.double().norm()), float(sd[k].grad.double().norm())
        e_norm = max(e_norm, abs(a - b) / max(b, 1e-5 * scale))
        assert abs(a - b) <= 2e-2 * max(b, 1e-5 * scale), (k, a, b)
    got = {k: p.grad for k, p in net.named_parameters()}
    ref = {k: sd[k].grad for k in got}
    worst, cos_min, n_ok, n_t, worst_k = grad_compare(got, ref)
    top2 = outs[0].detach().topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-4
    flips = int(((res[0].detach().cpu().argmax(1) != outs[0].detach().argmax(1)) & decided).sum())
    print(f"fp32 engine vs oracle: logits rel {e_logits:.2e}, |dloss| {abs(float(loss) - float(loss_ref)):.1e}, argmax flips outside ties "
          f"{flips}, gradients: worst element-wise rel {worst:.2e} ({worst_k}), lowest cosine {cos_min:.6f}, {n_ok}/{n_t} tensors "
          f"within 1e-3, worst norm rel {e_norm:.2e}")
    if tag:
        record_parity(tag, dict(dtype="fp32", logits_rel=e_logits, loss_abs=abs(float(loss) - float(loss_ref)), argmax_mismatch=flips,
                                grad_rel_worst=worst, grad_rel_worst_tensor=str(worst_k), grad_cos_min=cos_min,
                                grad_tensors_within_1e3=n_ok, grad_tensors=n_t, grad_norm_rel_worst=e_norm))
    assert flips == 0
    assert cos_min >= 0.999, (cos_min, worst_k)
    if f64:
        from tests.util import f64_bar
        sd64 = {k: v.detach().cpu().clone().double().requires_grad_(v.is_floating_point()) if v.is_floating_point() else v.detach().cpu().clone()
                for k, v in net.state_dict().items()}
        outs64 = oracle_forward(sd64, x.double())
        outs64 = outs64 if isinstance(outs64, (list, tuple)) else [outs64]
        (sum(ce_dice_loss(o, lab, w.double()) for o in outs64) / len(outs64)).backward()
        ratio, rk, e_eng, e_o32, amax, _ = f64_bar(got, ref, {k: sd64[k].grad for k in got})
        print(f"float64 bar: worst L2 err(engine) / (4 err(fp32 oracle) + floor) = {ratio:.2f} ({rk}); largest distance from the float64 "
              f"gradient: engine {e_eng:.2e}, fp32 oracle {e_o32:.2e} (of the tensor's norm); same ratio on the largest entry: {amax:.2f}")
        if tag:
            record_parity(tag + "_f64", dict(dtype="fp32", f64_ratio_worst=ratio, f64_ratio_worst_tensor=str(rk), f64_err_engine=e_eng,
                                             f64_err_oracle32=e_o32, f64_maxabs_ratio_worst=amax))
        assert ratio <= 1.0, (ratio, rk, e_eng, e_o32)
    else:
        assert worst <= grad_tol, (worst, worst_k)


def _blocky(classes, shape, batch, seed):
    g = torch.Generator().manual_seed(seed)
    x_shape = (batch,) + shape
    coarse = torch.randint(0, classes, (batch, 1) + tuple(max(1, s // 4) for s in shape[1:]), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=shape[1:], mode="nearest").long()
    x = torch.randn(x_shape, generator=g).clamp_(-7.4, 2.2)
    w = torch.ones(classes)
    w[0] = 0.5
    return x, lab, w


def test_resunet_ragged_batch2_matches_oracle(dev):
    from functools import partial
    from cbim_amd.model.dim3 import UNet
    from oracle.unet_ref import unet_forward
    torch.manual_seed(11)
    ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
    net = UNet(2, 8, scale=sc, kernel_size=ks, num_classes=5, block="BasicBlock", norm="in").to(dev)
    x, lab, w = _blocky(5, (2, 36, 52, 44), 2, 12)      # odd sizes at every pooling level, ragged 8x8x8 tiles
    _oracle_vs_engine(dev, net, partial(unet_forward, scale=sc, kernel_size=ks, block="BasicBlock"), x, lab, w, tag="resunet_ragged_b2_fp32")


def test_medformer_noncubic_batch2_matches_oracle(dev):
    from functools import partial
    from cbim_amd.model.dim3 import MedFormer
    from oracle.medformer_ref import medformer_forward
    from tests.medformer_checks import TINY
    torch.manual_seed(13)
    net = MedFormer(1, 4, **TINY).to(dev)
    x, lab, w = _blocky(4, (1, 48, 32, 80), 2, 14)
    fwd = partial(medformer_forward, map_size=TINY["map_size"], num_heads=TINY["num_heads"], fusion_heads=TINY["fusion_heads"],
                  fusion_depth=TINY["fusion_depth"], kernel_size=TINY["kernel_size"], scale=TINY["scale"], act="relu",
                  aux_loss=True)
    _oracle_vs_engine(dev, net, fwd, x, lab, w, tag="medformer_noncubic_b2_fp32")


def test_swin_unetr_noncubic_batch2_matches_oracle(dev):
    from cbim_amd.model.dim3 import SwinUNETR
    from oracle.swin_unetr_ref import swin_unetr_forward
    torch.manual_seed(15)
    net = SwinUNETR((32, 64, 96), 2, 3, feature_size=24).to(dev)
    x, lab, w = _blocky(3, (2, 32, 64, 96), 2, 16)
    _oracle_vs_engine(dev, net, swin_unetr_forward, x, lab, w, tag="swin_noncubic_b2_fp32")


def test_unetpp_matches_reference_golden(dev):
    from tests.unetpp_checks import assert_fp32, run
    print(assert_fp32(dev))
    r = run(dev, "bf16")
    assert r["logits_err"] < 0.25 and r["ce_err"] < 0.05 and r["dice_err"] < 0.02, r      # (absolute backstop)
    # round 6: the computed envelope against oracle/unet_ref.unetpp_forward (fp32 and under CPU autocast(bf16))
    from functools import partial
    from cbim_amd.model.dim3 import UNetPlusPlus
    from oracle.unet_ref import unetpp_forward
    from tests.unetpp_checks import KS, SCALE
    from tests.util import bf16_envelope_vs_oracle, load_golden
    g = load_golden("unetpp_b8_acdc")
    torch.manual_seed(int(g["seed"]))
    net = UNetPlusPlus(1, 8, scale=SCALE, kernel_size=KS, num_classes=4, block="BasicBlock", norm="in").to(dev)
    env, bad = bf16_envelope_vs_oracle(dev, net, partial(unetpp_forward, scale=SCALE, kernel_size=KS, block="BasicBlock"),
                                       torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"]),
                                       tag="unetpp_b8_acdc_bf16_envelope")
    assert not bad, bad


def test_attention_unet_matches_reference_golden(dev):
    from tests.attunet_checks import assert_fp32, run
    print(assert_fp32(dev, optimizer_step=True))
    # the base_chan-8 fixture has a 4-channel gate (out_ch // 2) — below the 8-channel bf16 chunk; bf16 mode must
    # refuse it loudly (shipped configs use base_chan 32 -> 16-channel gates)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        run(dev, "bf16")
