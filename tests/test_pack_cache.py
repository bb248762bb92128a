"""Packed-weight cache (ops.PackedWeights): one re-layout launch per optimizer step, keyed on Tensor._version.
Runs on the host-side kernel executor (not gpu)."""
import torch

import cbim_amd
from cbim_amd import ops
from cbim_amd.model.dim3 import UNet
from cbim_amd.training.optim import FusedAdamW


def _net():
    torch.manual_seed(3)
    return UNet(1, 4, scale=[[1, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=3, block="BasicBlock", norm="in")


def test_one_table_launch_per_optimizer_step_and_fresh_weights(dev, monkeypatch):
    cbim_amd.set_compute_dtype("fp32")
    try:
        ops.PACKED.clear()
        net = _net().to(dev)
        x = torch.randn(1, 1, 2, 16, 16, generator=torch.Generator().manual_seed(1)).to(dev)
        launches = []
        orig, orig_one = ops.PackedWeights._repack_all, ops.PackedWeights._pack_one
        monkeypatch.setattr(ops.PackedWeights, "_repack_all", lambda self: (launches.append(False), orig(self))[1])
        monkeypatch.setattr(ops.PackedWeights, "_pack_one", lambda self, e: (launches.append(True), orig_one(self, e))[1])
        opt = FusedAdamW(net.parameters(), lr=1e-2)
        net(x).sum().backward()
        n_first = len(launches)
        assert n_first == len(ops.PACKED.entries) > 20          # first step: one launch per new weight
        opt.step()
        opt.zero_grad()
        y1 = net(x)
        assert len(launches) == n_first + 1 and launches[-1] is False   # ONE launch re-packs every weight
        y1b = net(x)
        assert len(launches) == n_first + 1                     # unchanged weights: no launch
        assert torch.equal(y1, y1b)
        # the cached layouts are those of the UPDATED weights: same result as a cold cache
        ops.PACKED.clear()
        y2 = net(x)
        assert torch.equal(y1, y2)
        # in-place edits through torch (load_state_dict, manual surgery) are seen as well
        with torch.no_grad():
            net.inc.conv2.conv1.conv.weight.mul_(1.5)
        y3 = net(x)
        ops.PACKED.clear()
        assert torch.equal(y3, net(x)) and not torch.equal(y3, y2)
    finally:
        cbim_amd.set_compute_dtype(None)
        ops.PACKED.clear()


def test_dead_models_leave_the_table(dev, monkeypatch):
    cbim_amd.set_compute_dtype("fp32")
    monkeypatch.setattr(ops.PackedWeights, "MAX_IDLE", 2)
    try:
        ops.PACKED.clear()
        x = torch.randn(1, 1, 2, 16, 16).to(dev)
        old = _net().to(dev)
        old(x)
        n_one = len(ops.PACKED.entries)
        del old
        net = _net().to(dev)
        with torch.no_grad():
            for _ in range(5):       # 5 "optimizer steps" (version bumps) during which only `net` is used
                torch.autograd.graph.increment_version(list(net.parameters()))
                net(x)
        assert len(ops.PACKED.entries) == n_one                 # the first model's weights were evicted
    finally:
        cbim_amd.set_compute_dtype(None)
        ops.PACKED.clear()


def test_ema_update_invalidates_the_ema_net_packing(dev):
    """update_ema_variables writes the EMA parameters through raw pointers (cbim_ema_step): their version counters must
    move, or an ema_net forward after EMA updates runs on stale packed conv weights."""
    import copy
    from cbim_amd.training.utils import update_ema_variables
    cbim_amd.set_compute_dtype("fp32")
    try:
        ops.PACKED.clear()
        net = _net().to(dev)
        ema = copy.deepcopy(net)
        x = torch.randn(1, 1, 2, 16, 16, generator=torch.Generator().manual_seed(1)).to(dev)
        with torch.no_grad():
            y0 = ema(x)                                   # packs the EMA weights
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
            v0 = [p._version for p in ema.parameters()]
            for step in range(3):
                update_ema_variables(net, ema, 0.5, step + 1)
            assert all(p._version > v for p, v in zip(ema.parameters(), v0))
            y1 = ema(x)
            ops.PACKED.clear()
            y2 = ema(x)                                   # cold cache
        assert torch.equal(y1, y2) and not torch.equal(y0, y1)
    finally:
        cbim_amd.set_compute_dtype(None)
        ops.PACKED.clear()


def test_no_grad_forward_packs_no_dgrad_layout(dev):
    """sliding-window inference / validation run under torch.no_grad(): only the forward weight layout is packed."""
    cbim_amd.set_compute_dtype("fp32")
    try:
        ops.PACKED.clear()
        net = _net().to(dev)
        x = torch.randn(1, 1, 2, 16, 16).to(dev)
        with torch.no_grad():
            net(x)
        assert ops.PACKED.entries and all(e.p1 is None for e in ops.PACKED.entries.values())
        net(x).sum().backward()                           # training afterwards adds the dgrad layouts
        assert any(e.p1 is not None for e in ops.PACKED.entries.values())
    finally:
        cbim_amd.set_compute_dtype(None)
        ops.PACKED.clear()
