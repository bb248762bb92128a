"""FusedAdamW (+EMA) against torch.optim.AdamW and the reference's update_ema_variables loop
(/root/reference/training/utils.py:14,98-105) — shared by the CPU (host-side executor) and -m gpu suites."""
import copy

import torch
import torch.nn as nn

from cbim_amd.training.optim import FusedAdamW


def _ema_ref(model, ema_model, alpha, global_step):     # restatement of training/utils.py:98-105
    alpha = min(1 - 1 / (global_step + 1), alpha)
    for e, p in zip(ema_model.parameters(), model.parameters()):
        e.data.mul_(alpha).add_(p.data, alpha=1 - alpha)


def check_adamw_ema(dev, steps=4, seed=31):
    torch.manual_seed(seed)
    net = nn.Sequential(nn.Conv3d(3, 5, 3), nn.Conv3d(5, 7, 1), nn.Linear(11, 4097)).to(dev)   # 4097*11 > one chunk
    ref = copy.deepcopy(net)
    ema, ema_ref = copy.deepcopy(net), copy.deepcopy(net)
    kw = dict(lr=6e-4, betas=(0.9, 0.999), weight_decay=0.05, eps=1e-5)
    opt = FusedAdamW(net.parameters(), ema_model=ema, ema_alpha=0.99, **kw)
    opt_ref = torch.optim.AdamW(ref.parameters(), **kw)
    for step in range(steps):
        if step == 2:                                   # scheduler changes the lr (training/utils.py:51-95)
            for o in (opt, opt_ref):
                o.param_groups[0]["lr"] = 3e-4
        if step == 3:                                   # live edits of the other hyper-parameters reach the device too
            for o in (opt, opt_ref):
                o.param_groups[0]["weight_decay"] = 0.01
                o.param_groups[0]["betas"] = (0.8, 0.99)
                o.param_groups[0]["eps"] = 1e-6
        for p, q in zip(net.parameters(), ref.parameters()):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        opt.step()
        opt_ref.step()
        _ema_ref(ref, ema_ref, 0.99, step)
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-8, (step, n)
        for e, f in zip(ema.parameters(), ema_ref.parameters()):
            assert float((e - f).abs().max()) <= 2e-6 * float(f.abs().max()) + 1e-8, step
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == steps and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ref_sd = opt_ref.state_dict()
    for k in ("exp_avg", "exp_avg_sq"):
        a, b = sd["state"][2][k], ref_sd["state"][2][k]
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-10


def check_ema_buffers(dev):
    """An EMA model with buffers (SwinUNETR's relative_position_index) must be told where they come from."""
    net = nn.Sequential(nn.Conv3d(1, 2, 1), nn.BatchNorm3d(2)).to(dev)
    ema = copy.deepcopy(net)
    opt = FusedAdamW(net.parameters(), ema_model=ema, lr=1e-3)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    try:
        opt.step()
        raise AssertionError("FusedAdamW accepted an EMA model with buffers and no attach_buffers()")
    except RuntimeError as e:
        assert "attach_buffers" in str(e)
    opt.attach_buffers(net)
    with torch.no_grad():
        net[1].running_mean.fill_(3.0)
    opt.step()
    assert float(ema[1].running_mean[0]) == 3.0


def check_training_utils_surface(dev, seed=33):
    """cbim_amd.training.utils — the names train.py:16-22 imports — against the reference's own functions: EMA through
    cbim_ema_step vs the restated loop, the two lr schedules vs values printed by the REAL reference
    (/root/reference/training/utils.py:50-94; init_lr 6e-4, warmup 5, max_epoch 200, decay at [100, 150])."""
    import argparse
    from cbim_amd.training import utils as tu
    torch.manual_seed(seed)
    net = nn.Sequential(nn.Conv3d(2, 4, 3), nn.Linear(9, 4100)).to(dev)
    ema, ema_ref = copy.deepcopy(net), copy.deepcopy(net)
    for step in (0, 1, 2, 150):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p))
        tu.update_ema_variables(net, ema, 0.99, step)
        _ema_ref(net, ema_ref, 0.99, step)
        for e, f in zip(ema.parameters(), ema_ref.parameters()):
            # same roundings as mul_().add_(alpha=) where ATen contracts the add into an FMA (bit-equal on the host-side
            # executor); 2 ulp allowed in case a build does not contract
            assert float((e - f).abs().max()) <= 2.5e-7 * float(f.abs().max()), step
    args = argparse.Namespace(optimizer="adamw", base_lr=6e-4, betas=[0.9, 0.999], weight_decay=0.05, momentum=0.9)
    opt = tu.get_optimizer(args, net)
    assert isinstance(opt, FusedAdamW) and opt.param_groups[0]["eps"] == 1e-5 and opt.param_groups[0]["lr"] == 6e-4
    args.optimizer = "sgd"
    assert isinstance(tu.get_optimizer(args, net), torch.optim.SGD)
    exp_ref = [2.7268e-08, 2.01445e-07, 1.0993942e-05, 0.0006, 0.000583775428, 0.000463133704, 5.095939e-06]
    for e, want in zip((0, 1, 3, 5, 6, 50, 199), exp_ref):
        got = tu.exp_lr_scheduler_with_warmup(opt, 6e-4, e, 5, 200)
        assert abs(got - want) <= 1e-6 * want + 1e-12 and opt.param_groups[0]["lr"] == got, (e, got, want)
    opt.param_groups[0]["lr"] = 6e-4
    ms_ref = [2.7268e-08, 1.488177e-06, 0.0006, 0.0006, 6e-05, 6e-05, 6e-06, 6e-06]
    for e, want in zip((0, 2, 5, 6, 100, 101, 150, 151), ms_ref):
        got = tu.multistep_lr_scheduler_with_warmup(opt, 6e-4, e, 5, [100, 150], 200)
        assert abs(got - want) <= 1e-6 * want + 1e-12, (e, got, want)
