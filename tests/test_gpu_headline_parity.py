"""-m gpu: parity ON the headline configuration (BASELINE.json configs[1]: 3D UNet ResBasicBlock, base 32, 16 classes,
1x1x128^3) against the CPU oracle (oracle/unet_ref.py, pinned to the real reference by tests/golden):

  * fp32 engine mode: logits within 1e-3 (relative to the logit range), argmax label maps identical outside fp32 ties
    (top-2 gap < 1e-4), hard Dice within 0.002 — the bars of BASELINE.json's north_star, on the benchmarked shape;
  * bf16 engine mode (the benchmarked dtype) on TRAINED weights: ~200 AdamW steps on a learnable synthetic volume, then the
    engine's hard Dice against the labels must be within 0.002 of the fp32 oracle's on the same weights and input
    (SURVEY.md §8d ii: with trained weights the class margins are real, so Dice is a meaningful bar for bf16).
"""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

KS, SC = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
BASE, CLASSES, SIZE = 32, 16, 128


def _volume(seed, informative):
    """128^3 label map of 16^3 blocks; the image is either clamped noise (bench.py's synthetic input) or a noisy
    class-dependent intensity (learnable in a few hundred steps)."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, CLASSES, (1, 1, SIZE // 16, SIZE // 16, SIZE // 16), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=(SIZE,) * 3, mode="nearest").long()
    noise = torch.randn(1, 1, SIZE, SIZE, SIZE, generator=g)
    if informative:
        x = torch.linspace(-3.0, 2.0, CLASSES)[lab] + 0.35 * noise
    else:
        x = noise.clamp_(-7.4, 2.2)
    return x, lab


def _oracle_logits(sd, x):
    from oracle import unet_ref
    with torch.no_grad():
        t0 = time.perf_counter()
        lo = unet_ref.unet_forward({k: v.float().cpu() for k, v in sd.items()}, x.cpu(), scale=SC, kernel_size=KS,
                                   block="BasicBlock")
        print(f"oracle forward at {SIZE}^3: {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads")
    return lo


def test_headline_config_fp32_engine_matches_oracle_at_128(dev):
    import cbim_amd
    from cbim_amd.model.dim3 import UNet
    from oracle import loss_ref, unet_ref
    x, lab = _volume(11, informative=False)
    sd = unet_ref.make_unet_state_dict(1, BASE, CLASSES, KS, "BasicBlock", seed=2023)
    lo = _oracle_logits(sd, x)
    cbim_amd.set_compute_dtype("fp32")
    try:
        net = UNet(1, BASE, scale=SC, kernel_size=KS, num_classes=CLASSES, block="BasicBlock", norm="in").to(dev)
        net.load_state_dict(sd)
        with torch.no_grad():
            lg = net(x.to(dev)).cpu()
    finally:
        cbim_amd.set_compute_dtype(None)
    err = float((lg - lo).abs().max() / lo.abs().max())
    top2 = lo.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-4
    diff = lg.argmax(1) != lo.argmax(1)
    d_o = loss_ref.hard_dice(lo.argmax(1), lab.squeeze(1), CLASSES)
    d_e = loss_ref.hard_dice(lg.argmax(1), lab.squeeze(1), CLASSES)
    ddice = float((d_o - d_e).abs().max())
    print(f"128^3 base 32 fp32 engine vs oracle: logits rel err {err:.2e}, argmax mismatches {int(diff.sum())} "
          f"({int((diff & decided).sum())} outside fp32 ties, {int((~decided).sum())} tie voxels), max |dDice| {ddice:.2e}")
    assert err < 1e-3
    assert int((diff & decided).sum()) == 0 and int(diff.sum()) <= 1e-4 * diff.numel()
    assert ddice <= 0.002


def test_headline_config_bf16_dice_within_0p002_of_oracle_on_trained_weights(dev):
    import cbim_amd
    from cbim_amd.model.dim3 import UNet
    from cbim_amd.training.losses import DiceCELoss
    from cbim_amd.training.optim import FusedAdamW
    from oracle import loss_ref
    x, lab = _volume(12, informative=True)
    xd, ld = x.to(dev), lab.to(dev)
    cbim_amd.set_compute_dtype("bf16")
    try:
        torch.manual_seed(2023)
        net = UNet(1, BASE, scale=SC, kernel_size=KS, num_classes=CLASSES, block="BasicBlock", norm="in").to(dev)
        w = torch.ones(CLASSES, device=dev)
        crit = DiceCELoss(w).to(dev)
        opt = FusedAdamW(net.parameters(), lr=2e-3, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-5)
        losses = []
        for i in range(200):
            opt.zero_grad(set_to_none=True)
            loss = crit(net(xd), ld)
            loss.backward()
            opt.step()
            if i % 40 == 0 or i == 199:
                losses.append(float(loss))
        with torch.no_grad():
            lg = net(xd).float().cpu()
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    finally:
        cbim_amd.set_compute_dtype(None)
    lo = _oracle_logits(sd, x)
    tgt = lab.squeeze(1)
    d_o = loss_ref.hard_dice(lo.argmax(1), tgt, CLASSES)
    d_e = loss_ref.hard_dice(lg.argmax(1), tgt, CLASSES)
    agree = float((lg.argmax(1) == lo.argmax(1)).float().mean())
    ddice = float((d_o - d_e).abs().max())
    print(f"200 bf16 AdamW steps, loss {losses}; oracle mean Dice {float(d_o.mean()):.4f}, engine {float(d_e.mean()):.4f}, "
          f"max per-class |dDice| {ddice:.2e}, argmax agreement {agree:.5f}")
    assert losses[-1] < 0.5 * losses[0]                  # it trained
    assert float(d_o.mean()) > 0.5                       # the margins are real: Dice is a meaningful bar here
    assert ddice <= 0.002
    assert agree > 0.995
