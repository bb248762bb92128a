"""AttentionUNet whole-model parity against the golden of the REAL reference (tests/golden/make_golden_attunet.py)."""
import numpy as np
import torch

import cbim_amd
from cbim_amd import functional as Fn
from cbim_amd.model.dim3 import AttentionUNet
from tests.util import load_golden, rel_err

SCALE = [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
KS = [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]


def build():
    from oracle.unet_ref import state_dict_checksum
    g = load_golden("attunet_b8")
    torch.manual_seed(int(g["seed"]))
    net = AttentionUNet(1, 8, scale=SCALE, kernel_size=KS, num_classes=4, block="BasicBlock", norm="in")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert abs(state_dict_checksum(sd) - float(g["sd_checksum"])) < 1e-6 and sum(p.numel() for p in net.parameters()) == int(g["n_params"])
    return net, g


def run(dev, mode="fp32", optimizer_step=False):
    net, g = build()
    net = net.to(dev)
    cbim_amd.set_compute_dtype(mode)
    try:
        logits = net(torch.from_numpy(g["x"]).to(dev))
        both = Fn.DiceCEFn.apply(logits, torch.from_numpy(g["label"]).to(dev), torch.from_numpy(g["weight"]).to(dev))
        both[2].backward()
    finally:
        cbim_amd.set_compute_dtype(None)
    params = dict(net.named_parameters())
    keys = [str(k) for k in g["keys"]]
    scale = float(g["grad_norms"].max())
    errs = []
    for k, ref in zip(keys, g["grad_norms"]):
        if ref < 0:                                   # conv_ch: unused by the reference -> no gradient
            assert params[k].grad is None, k
            continue
        errs.append(abs(float(params[k].grad.double().norm()) - ref) / max(ref, 1e-6 * scale))
    res = {
        "logits_err": rel_err(logits.detach().cpu(), g["logits"]),
        "argmax_mismatch": int((logits.argmax(1).cpu() != torch.from_numpy(g["logits"]).argmax(1)).sum()),
        "ce_err": abs(float(both[0]) - float(g["ce"])), "dice_err": abs(float(both[1]) - float(g["dice"])),
        "grad_norm_err": max(errs),
        "g_first": rel_err(params["inc.conv1.weight"].grad.cpu(), g["g:inc.conv1.weight"]),
        "g_head": rel_err(params["outc.weight"].grad.cpu(), g["g:outc.weight"]),
    }
    if optimizer_step:                                # parameters without gradient are skipped like torch.optim does
        from cbim_amd.training.optim import FusedAdamW
        before = params["up1.conv_ch.weight"].detach().clone()
        opt = FusedAdamW(net.parameters(), lr=1e-3)
        opt.step()
        assert torch.equal(before, params["up1.conv_ch.weight"].detach())
    return res


def assert_fp32(dev, optimizer_step=False):
    r = run(dev, optimizer_step=optimizer_step)
    assert r["logits_err"] < 1e-3 and r["argmax_mismatch"] == 0, r
    assert r["ce_err"] < 1e-4 and r["dice_err"] < 1e-4, r
    assert r["grad_norm_err"] < 1e-2 and r["g_first"] < 2e-2 and r["g_head"] < 1e-3, r
    return r
