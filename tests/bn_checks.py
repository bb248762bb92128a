"""`norm: bn` and `pool=False` whole-model parity against the goldens of tests/golden/make_golden_bn.py (the REAL reference's
UNet with nn.BatchNorm3d / with strided first blocks, one training step + the eval-mode forward) — shared by the CPU (host-side
executor) and -m gpu suites."""
import torch

import cbim_amd
from cbim_amd import functional as Fn
from cbim_amd.model.dim3 import UNet
from tests.test_oracle import BN_CASES, bn_state_dict
from tests.util import rel_err


def run_case(name, dev, mode):
    in_ch, base, classes, scale, ks, block, seed, norm, pool = BN_CASES[name]
    sd, g = bn_state_dict(name)
    cbim_amd.set_compute_dtype(mode)
    try:
        net = UNet(in_ch, base, scale=scale, kernel_size=ks, num_classes=classes, block=block, norm=norm, pool=pool)
        assert list(net.state_dict().keys()) == list(sd.keys())          # the reference's state_dict layout, BatchNorm buffers included
        net.load_state_dict(sd)
        net = net.to(dev).train()
        x, lab, w = (torch.from_numpy(g[k]).to(dev) for k in ("x", "label", "weight"))
        logits = net(x)
        both = Fn.DiceCEFn.apply(logits, lab, w)
        both[2].backward()
        params = dict(net.named_parameters())
        pk = [str(k) for k in g["param_keys"]]
        res = {"logits_err": rel_err(logits.detach().float().cpu(), g["logits"]), "ce": float(both[0]), "dice": float(both[1]),
               "grad_norm_err": max(abs(float(params[k].grad.double().norm()) - b) / max(b, 1e-6) for k, b in zip(pk, g["grad_norms"])),
               "grad_full_err": max(rel_err(params[k[2:]].grad.cpu(), g[k]) for k in g.files if k.startswith("g:")),
               "running_err": max([rel_err(net.state_dict()[k[2:]].double().cpu(), g[k].astype("float64")) for k in g.files if k.startswith("r:")] + [0.0])}
        net.eval()
        with torch.no_grad():
            res["eval_logits_err"] = rel_err(net(x).float().cpu(), g["logits_eval"])
        return res, g
    finally:
        cbim_amd.set_compute_dtype(None)


def assert_fp32(name, dev):
    r, g = run_case(name, dev, "fp32")
    assert r["logits_err"] < 1e-3 and r["eval_logits_err"] < 1e-3, r
    assert abs(r["ce"] - float(g["ce"])) < 1e-4 and abs(r["dice"] - float(g["dice"])) < 1e-4, r
    assert r["grad_norm_err"] < 1e-2 and r["grad_full_err"] < 2e-2 and r["running_err"] < 1e-4, r
    return r
