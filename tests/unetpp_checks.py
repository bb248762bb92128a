"""UNet++ whole-model parity against the golden of the REAL reference (tests/golden/make_golden_unetpp.py)."""
import numpy as np
import torch

import cbim_amd
from cbim_amd import functional as Fn
from cbim_amd.model.dim3 import UNetPlusPlus
from tests.util import load_golden, rel_err

SCALE = [[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]]
KS = [[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]


def run(dev, mode="fp32", backward=True):
    from oracle.unet_ref import state_dict_checksum
    g = load_golden("unetpp_b8_acdc")
    torch.manual_seed(int(g["seed"]))
    net = UNetPlusPlus(1, 8, scale=SCALE, kernel_size=KS, num_classes=4, block="BasicBlock", norm="in")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert abs(state_dict_checksum(sd) - float(g["sd_checksum"])) < 1e-6 and sum(p.numel() for p in net.parameters()) == int(g["n_params"])
    net = net.to(dev)
    cbim_amd.set_compute_dtype(mode)
    try:
        logits = net(torch.from_numpy(g["x"]).to(dev))
        both = Fn.DiceCEFn.apply(logits, torch.from_numpy(g["label"]).to(dev), torch.from_numpy(g["weight"]).to(dev))
        if backward:
            both[2].backward()
    finally:
        cbim_amd.set_compute_dtype(None)
    res = {
        "logits_err": rel_err(logits.detach().cpu(), g["logits"]),
        "argmax_mismatch": int((logits.argmax(1).cpu() != torch.from_numpy(g["logits"]).argmax(1)).sum()),
        "ce_err": abs(float(both[0]) - float(g["ce"])), "dice_err": abs(float(both[1]) - float(g["dice"])),
    }
    if not backward:
        return res
    params = dict(net.named_parameters())
    keys = [str(k) for k in g["keys"]]
    gn = np.array([float(params[k].grad.double().norm()) for k in keys])
    scale = float(g["grad_norms"].max())
    return {
        "logits_err": rel_err(logits.detach().cpu(), g["logits"]),
        "argmax_mismatch": int((logits.argmax(1).cpu() != torch.from_numpy(g["logits"]).argmax(1)).sum()),
        "ce_err": abs(float(both[0]) - float(g["ce"])), "dice_err": abs(float(both[1]) - float(g["dice"])),
        "grad_norm_err": float(np.max(np.abs(gn - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-6 * scale))),
        "g_first": rel_err(params["conv0_0.0.conv1.conv.weight"].grad.cpu(), g["g:conv0_0.0.conv1.conv.weight"]),
        "g_head": rel_err(params["output.weight"].grad.cpu(), g["g:output.weight"]),
    }


def assert_fp32(dev, backward=True):
    r = run(dev, backward=backward)
    assert r["logits_err"] < 1e-3 and r["argmax_mismatch"] == 0, r
    assert r["ce_err"] < 1e-4 and r["dice_err"] < 1e-4, r
    if backward:
        assert r["grad_norm_err"] < 1e-2 and r["g_first"] < 2e-2 and r["g_head"] < 1e-3, r
    return r


# ---- `norm: bn | ln` (round 5): tests/norm_branch_checks.py holds the shared code -------------------------------------------------
from tests.norm_branch_checks import NORM_CASES, assert_norm_fp32, build_norm_case, run_norm_case  # noqa: E402,F401
