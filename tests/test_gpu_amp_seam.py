"""-m gpu: the AMP seam the reference's trainer drives (/root/reference/train.py:188-203, train_ddp.py:181): the engine
module entered under torch.autocast(device_type='cuda', dtype=torch.float16) with a GradScaler, backward() and
scaler.step() INSIDE the autocast block, CrossEntropyLoss + the reference-style DiceLoss call shapes.  Under autocast the
engine computes in bf16 storage (functional.compute_dtype), logits and losses stay fp32; the scaled gradients must come
out finite (no skipped step) and, once unscaled, equal the gradients of the same bf16 step without a scaler."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev):
    import cbim_amd
    from cbim_amd.model.dim3 import UNet
    from cbim_amd.training.losses import DiceLoss
    torch.manual_seed(11)
    ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
    net = UNet(1, 8, scale=sc, kernel_size=ks, num_classes=4, block="BasicBlock", norm="in").to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1, 32, 32, 32, generator=g).to(dev)
    lab = torch.randint(0, 4, (1, 1, 32, 32, 32), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.5, 1.0, 1.0, 1.0]).to(dev))
    return net, x, lab, crit, DiceLoss()


def test_autocast_fp16_with_gradscaler_as_train_py_does(dev):
    import cbim_amd
    assert dev == "cuda"
    cbim_amd.set_compute_dtype(None)                   # the dtype follows autocast, as in the reference's --amp run
    d = torch.device("cuda", 0)
    net, x, lab, criterion, criterion_dl = _setup(d)
    optimizer = torch.optim.AdamW(net.parameters(), lr=6e-4, betas=(0.9, 0.999), weight_decay=0.05, eps=1e-5)
    scaler = torch.amp.GradScaler("cuda")
    w0 = [p.detach().clone() for p in net.parameters()]
    # --- train.py:188-203, line for line -------------------------------------------------------------------------
    optimizer.zero_grad()
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        result = net(x)
        loss = criterion(result, lab.squeeze(1)) + criterion_dl(result, lab)
        scaler.scale(loss).backward()
        scaled = [p.grad.detach().clone() for p in net.parameters()]
        scaler.step(optimizer)
        scaler.update()
    torch.cuda.synchronize()
    assert result.dtype == torch.float32 and torch.isfinite(loss)
    assert scaler.get_scale() == 65536.0              # no inf/nan was found: the step was taken, the scale kept
    assert any(not torch.equal(p.detach(), w) for p, w in zip(net.parameters(), w0))
    # --- the same step in bf16 engine mode without autocast / scaler, from the same weights ------------------------
    with torch.no_grad():
        for p, w in zip(net.parameters(), w0):
            p.copy_(w)
    cbim_amd.set_compute_dtype("bf16")
    try:
        net.zero_grad(set_to_none=True)
        out2 = net(x)
        loss2 = criterion(out2, lab.squeeze(1)) + criterion_dl(out2, lab)
        loss2.backward()
    finally:
        cbim_amd.set_compute_dtype(None)
    assert abs(float(loss) - float(loss2)) <= 1e-5 * max(1.0, abs(float(loss2)))
    for p, gs in zip(net.parameters(), scaled):
        assert torch.isfinite(gs).all()
        ref = p.grad
        # a 2^16-scaled gradient runs through the same bf16 kernels: scaling by a power of two commutes with every
        # rounding on the way except where values leave the normal range (none here)
        err = float((gs / 65536.0 - ref).abs().max())
        assert err <= 2e-2 * float(ref.abs().max()) + 1e-12, err
