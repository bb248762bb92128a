"""`norm: bn | ln` in UNet++ and AttentionUNet (round 5) against fixtures of the REAL reference (tests/golden/make_golden_unetpp.py
norms, make_golden_attunet.py norms): perturbed affine parameters, one training step (logits, CE, Dice, every gradient norm, full
first-layer / gate / head gradients, running statistics) and the eval-mode logits.  Shared by the CPU (host-side executor) and -m gpu
suites and by the oracle pins."""
import torch

import cbim_amd
from cbim_amd import functional as Fn
from tests.util import load_golden, rel_err

PP = dict(scale=[[1, 2, 2], [1, 2, 2], [2, 2, 2], [2, 2, 2]], kernel_size=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]])
ATT = dict(scale=[[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], kernel_size=[[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]])
NORM_CASES = {   # name: (model, block, norm)
    "unetpp_bn_b8": ("unetpp", "BasicBlock", "bn"), "unetpp_ln_b8": ("unetpp", "SingleConv", "ln"),
    "attunet_bn_b8": ("attunet", "SingleConv", "bn"), "attunet_ln_b8": ("attunet", "BasicBlock", "ln"),
}


def geometry(name):
    return PP if NORM_CASES[name][0] == "unetpp" else ATT


def build_norm_case(name):
    """the engine module with the fixture's weights: seeded constructor (same draws as the reference's) + the perturbed affine
    parameters, checksum-checked against the reference's state_dict"""
    from cbim_amd.model.dim3 import AttentionUNet, UNetPlusPlus
    from oracle.unet_ref import state_dict_checksum
    model, block, norm = NORM_CASES[name]
    g = load_golden(name)
    torch.manual_seed(int(g["seed"]))
    cls = UNetPlusPlus if model == "unetpp" else AttentionUNet
    net = cls(1, 8, num_classes=4, block=block, norm=norm, **geometry(name))
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    for k in g.files:
        if k.startswith("p:"):
            sd[k[2:]] = torch.from_numpy(g[k]).clone()
    net.load_state_dict(sd)
    chk = state_dict_checksum({k: v for k, v in net.state_dict().items() if v.is_floating_point()})
    assert abs(chk - float(g["sd_checksum"])) <= 1e-9 * max(1.0, abs(chk)), (chk, float(g["sd_checksum"]))
    return net, g


def run_norm_case(name, dev, mode):
    net, g = build_norm_case(name)
    cbim_amd.set_compute_dtype(mode)
    try:
        net = net.to(dev).train()
        x, lab, w = (torch.from_numpy(g[k]).to(dev) for k in ("x", "label", "weight"))
        logits = net(x)
        both = Fn.DiceCEFn.apply(logits, lab, w)
        both[2].backward()
        params = dict(net.named_parameters())
        pk = [str(k) for k in g["param_keys"]]
        scale = float(g["grad_norms"].max())
        errs = []
        for k, b in zip(pk, g["grad_norms"]):
            if b < 0:                                 # AttentionUNet's conv_ch: declared, unused by the reference -> no gradient
                assert params[k].grad is None, k
                continue
            errs.append(abs(float(params[k].grad.double().norm()) - b) / max(b, 1e-6 * scale))
        res = {"logits_err": rel_err(logits.detach().float().cpu(), g["logits"]), "ce": float(both[0]), "dice": float(both[1]),
               "grad_norm_err": max(errs),
               "grad_full_err": max(rel_err(params[k[2:]].grad.cpu(), g[k]) for k in g.files if k.startswith("g:")),
               "running_err": max([rel_err(net.state_dict()[k[2:]].double().cpu(), g[k].astype("float64")) for k in g.files if k.startswith("r:")] + [0.0])}
        net.eval()
        with torch.no_grad():
            res["eval_logits_err"] = rel_err(net(x).float().cpu(), g["logits_eval"])
        return res, g
    finally:
        cbim_amd.set_compute_dtype(None)


def assert_norm_fp32(name, dev):
    r, g = run_norm_case(name, dev, "fp32")
    assert r["logits_err"] < 1e-3 and r["eval_logits_err"] < 1e-3, r
    assert abs(r["ce"] - float(g["ce"])) < 1e-4 and abs(r["dice"] - float(g["dice"])) < 1e-4, r
    assert r["grad_norm_err"] < 1e-2 and r["grad_full_err"] < 2e-2 and r["running_err"] < 1e-4, r
    return r
