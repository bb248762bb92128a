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
