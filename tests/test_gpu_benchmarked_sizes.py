"""-m gpu: oracle parity AT THE BENCHMARKED SIZE for the two other headline configurations (BASELINE.json configs[2] and
configs[4]): MedFormer (amos_ct/medformer_3d.yaml, aux loss) on 1x1x128^3 and SwinUNETR (feature 48, 4 modalities, 4
classes) on 1x4x128^3.  fp32 engine mode against oracle/medformer_ref.py / oracle/swin_unetr_ref.py evaluated on the host
cores with the same weights: logits within 1e-3 of the logit range, loss within 1e-4, every parameter-gradient norm within
2 %.  Plus the trained-weights Dice bar of the bf16 mode for MedFormer and SwinUNETR (as tests/test_gpu_headline_parity.py
does for the ResUNet).  The 64^3 / tiny goldens of tests/golden come from the REAL reference; these tests carry the same comparison to the
shape bench.py times."""
import time
from functools import partial

import pytest
import torch

pytestmark = pytest.mark.gpu

MEDFORMER_AMOS = dict(base_chan=32, map_size=[4, 4, 4], conv_block="BasicBlock", conv_num=[2, 1, 0, 0, 0, 1, 2, 2],
                      trans_num=[0, 1, 4, 6, 4, 1, 0, 0], chan_num=[64, 128, 256, 320, 256, 128, 64, 32],
                      num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10,
                      expansion=4, attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu",
                      kernel_size=[[3, 3, 3]] * 5, scale=[[2, 2, 2]] * 4, aux_loss=True)  # config/amos_ct/medformer_3d.yaml
SIZE = 128


def _medformer_oracle():
    from oracle.medformer_ref import medformer_forward
    m = MEDFORMER_AMOS
    return partial(medformer_forward, map_size=m["map_size"], num_heads=m["num_heads"], fusion_heads=m["fusion_heads"],
                   fusion_depth=m["fusion_depth"], kernel_size=m["kernel_size"], scale=m["scale"], act="relu", aux_loss=True)


def _data(classes, in_ch, seed, informative=False):
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, classes, (1, 1, SIZE // 16, SIZE // 16, SIZE // 16), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=(SIZE,) * 3, mode="nearest").long()
    noise = torch.randn(1, in_ch, SIZE, SIZE, SIZE, generator=g)
    if informative:
        x = torch.linspace(-3.0, 2.0, classes)[lab].expand(1, in_ch, SIZE, SIZE, SIZE) + 0.35 * noise
    else:
        x = noise.clamp_(-7.4, 2.2)
    w = torch.ones(classes)
    w[0] = 0.5
    return x, lab, w


def test_medformer_amos_128_fp32_engine_matches_oracle(dev):
    from cbim_amd.model.dim3 import MedFormer
    from tests.test_gpu_parity import _oracle_vs_engine
    torch.manual_seed(2023)
    net = MedFormer(1, 16, **MEDFORMER_AMOS).to(dev)
    x, lab, w = _data(16, 1, 21)
    t0 = time.perf_counter()
    _oracle_vs_engine(dev, net, _medformer_oracle(), x, lab, w, tag="medformer_amos_128_fp32")
    print(f"MedFormer AMOS 1x1x{SIZE}^3: oracle + engine fwd/loss/bwd compared in {time.perf_counter() - t0:.0f} s")


def test_swin_unetr_4x128_fp32_engine_matches_oracle(dev):
    from cbim_amd.model.dim3 import SwinUNETR
    from oracle.swin_unetr_ref import swin_unetr_forward
    from tests.test_gpu_parity import _oracle_vs_engine
    torch.manual_seed(2023)
    net = SwinUNETR((SIZE,) * 3, 4, 4, feature_size=48).to(dev)
    x, lab, w = _data(4, 4, 22)
    t0 = time.perf_counter()
    _oracle_vs_engine(dev, net, swin_unetr_forward, x, lab, w, tag="swin_unetr_4x128_fp32")
    print(f"SwinUNETR 1x4x{SIZE}^3: oracle + engine fwd/loss/bwd compared in {time.perf_counter() - t0:.0f} s")


@pytest.mark.parametrize("model", ["resunet", "medformer", "swin_unetr"])
def test_benchmarked_architectures_64_fp32_engine_inside_the_f64_bar(dev, model):
    """The three benchmarked architectures (configs[1], [2], [4]: full widths, 16 / 16 / 4 classes) at 1xCx64^3, where the oracle
    can also be evaluated in FLOAT64 within seconds: every parameter gradient of the fp32 engine is at most 4x as far (L2) from
    the float64 gradient as the stock-torch fp32 evaluation of the same network is (tests.util.f64_bar: measured 3.0x / 2.3x / 0.5x;
    VERDICT r04 weak 1).
    The 128^3 tests above keep the fp32-vs-fp32 numbers (cosine, norms, logits, loss, argmax)."""
    from functools import partial
    from cbim_amd.model.dim3 import MedFormer, SwinUNETR, UNet
    from oracle.swin_unetr_ref import swin_unetr_forward
    from oracle.unet_ref import unet_forward
    from tests.test_gpu_parity import _oracle_vs_engine
    torch.manual_seed(2023)
    S = 64
    if model == "resunet":
        ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
        net, classes, in_ch = UNet(1, 32, scale=sc, kernel_size=ks, num_classes=16, block="BasicBlock", norm="in").to(dev), 16, 1
        fwd = partial(unet_forward, scale=sc, kernel_size=ks, block="BasicBlock")
    elif model == "medformer":
        net, classes, in_ch, fwd = MedFormer(1, 16, **MEDFORMER_AMOS).to(dev), 16, 1, _medformer_oracle()
    else:
        net, classes, in_ch, fwd = SwinUNETR((S,) * 3, 4, 4, feature_size=48).to(dev), 4, 4, swin_unetr_forward
    g = torch.Generator().manual_seed(31)
    coarse = torch.randint(0, classes, (1, 1, S // 8, S // 8, S // 8), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=(S,) * 3, mode="nearest").long()
    x = torch.randn(1, in_ch, S, S, S, generator=g).clamp_(-7.4, 2.2)
    w = torch.ones(classes)
    w[0] = 0.5
    t0 = time.perf_counter()
    _oracle_vs_engine(dev, net, fwd, x, lab, w, tag=f"{model}_64_fp32", f64=True)
    print(f"{model} 1x{in_ch}x{S}^3: fp32 + float64 oracle and engine compared in {time.perf_counter() - t0:.0f} s")


def test_medformer_amos_128_bf16_dice_within_0p002_of_oracle_on_trained_weights(dev):
    """~200 bf16 AdamW steps on a learnable synthetic volume (class margins become real), then the engine's hard Dice must
    be within 0.002 of the fp32 oracle's on the same weights and input (SURVEY.md §8d ii)."""
    import cbim_amd
    from cbim_amd.model.dim3 import MedFormer
    from cbim_amd.training.losses import DiceCELoss
    from cbim_amd.training.optim import FusedAdamW
    from oracle import loss_ref
    x, lab, _ = _data(16, 1, 23, informative=True)
    xd, ld = x.to(dev), lab.to(dev)
    cbim_amd.set_compute_dtype("bf16")
    try:
        torch.manual_seed(2023)
        net = MedFormer(1, 16, **MEDFORMER_AMOS).to(dev)
        crit = DiceCELoss(torch.ones(16, device=dev)).to(dev)
        opt = FusedAdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-5)
        losses = []
        for i in range(200):
            opt.zero_grad(set_to_none=True)
            out = net(xd)
            loss = sum(0.5 * crit(o, ld) for o in out)      # deep supervision, train.py:207-210
            loss.backward()
            opt.step()
            if i % 40 == 0 or i == 199:
                losses.append(float(loss))
        with torch.no_grad():
            lg = net(xd)[0].float().cpu()
        sd = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    finally:
        cbim_amd.set_compute_dtype(None)
    with torch.no_grad():
        t0 = time.perf_counter()
        lo = _medformer_oracle()(sd, x)[0]
        print(f"oracle forward at {SIZE}^3: {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads")
    tgt = lab.squeeze(1)
    d_o = loss_ref.hard_dice(lo.argmax(1), tgt, 16)
    d_e = loss_ref.hard_dice(lg.argmax(1), tgt, 16)
    agree = float((lg.argmax(1) == lo.argmax(1)).float().mean())
    ddice = float((d_o - d_e).abs().max())
    print(f"200 bf16 AdamW steps, loss {losses}; oracle mean Dice {float(d_o.mean()):.4f}, engine {float(d_e.mean()):.4f}, "
          f"max per-class |dDice| {ddice:.2e}, argmax agreement {agree:.5f}")
    assert losses[-1] < 0.5 * losses[0]
    assert float(d_o.mean()) > 0.5
    assert ddice <= 0.002
    assert agree > 0.995


def test_swin_unetr_4x128_bf16_dice_within_0p002_of_oracle_on_trained_weights(dev):
    """The same trained-weights bar for SwinUNETR (feature 48, 4 modalities, 4 classes) at the benchmarked 1x4x128^3."""
    import cbim_amd
    from cbim_amd.model.dim3 import SwinUNETR
    from cbim_amd.training.losses import DiceCELoss
    from cbim_amd.training.optim import FusedAdamW
    from oracle import loss_ref
    from oracle.swin_unetr_ref import swin_unetr_forward
    x, lab, _ = _data(4, 4, 24, informative=True)
    xd, ld = x.contiguous().to(dev), lab.to(dev)
    cbim_amd.set_compute_dtype("bf16")
    try:
        torch.manual_seed(2023)
        net = SwinUNETR((SIZE,) * 3, 4, 4, feature_size=48).to(dev)
        crit = DiceCELoss(torch.ones(4, device=dev)).to(dev)
        opt = FusedAdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-5)
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
        sd = {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}
    finally:
        cbim_amd.set_compute_dtype(None)
    with torch.no_grad():
        t0 = time.perf_counter()
        lo = swin_unetr_forward(sd, x)
        lo = lo[0] if isinstance(lo, (list, tuple)) else lo
        print(f"oracle forward at 4x{SIZE}^3: {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads")
    tgt = lab.squeeze(1)
    d_o = loss_ref.hard_dice(lo.argmax(1), tgt, 4)
    d_e = loss_ref.hard_dice(lg.argmax(1), tgt, 4)
    agree = float((lg.argmax(1) == lo.argmax(1)).float().mean())
    ddice = float((d_o - d_e).abs().max())
    print(f"200 bf16 AdamW steps, loss {losses}; oracle mean Dice {float(d_o.mean()):.4f}, engine {float(d_e.mean()):.4f}, "
          f"max per-class |dDice| {ddice:.2e}, argmax agreement {agree:.5f}")
    assert losses[-1] < 0.5 * losses[0]
    assert float(d_o.mean()) > 0.5
    assert ddice <= 0.002
    assert agree > 0.995
