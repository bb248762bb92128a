"""VNet parity checks against the golden fixture produced by the REAL reference in training mode with recorded Dropout3d masks
(tests/golden/make_golden_vnet.py) — shared by the CPU (oracle, host-side executor) and -m gpu suites."""
import torch

from tests.util import grad_compare, load_golden, rel_err

SCALE = [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
IN_CH, BASE, CLASSES = 1, 8, 4


def _masks(g):
    return [torch.from_numpy(g[f"mask{i}"]) for i in range(int(g["n_masks"]))]


def _golden_state_dict(g):
    """the reference constructor's weights from the fixture's seed (same module construction order = same RNG consumption),
    verified against the checksum recorded from the real reference"""
    from cbim_amd.model.dim3 import VNet
    from oracle.unet_ref import state_dict_checksum
    torch.manual_seed(int(g["seed"]))
    net = VNet(IN_CH, CLASSES, scale=SCALE, baseChans=BASE)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]], "state_dict layout differs from the reference's"
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    pk = [str(k) for k in g["param_keys"]]
    chk = state_dict_checksum({k: sd[k] for k in pk})
    assert abs(chk - float(g["sd_checksum"])) <= 1e-9 * max(1.0, abs(chk)), (chk, float(g["sd_checksum"]))
    return net, sd


def oracle_vs_golden():
    """oracle/vnet_ref.py against the real reference's outputs: logits, losses, every recorded gradient."""
    from oracle import loss_ref, vnet_ref
    g = load_golden("vnet_b8")
    _, sd = _golden_state_dict(g)
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    lo = vnet_ref.vnet_forward(sdr, x, SCALE, masks=_masks(g))
    loss = loss_ref.ce_dice_loss(lo, lab, w)
    loss.backward()
    res = {"logits": rel_err(lo.detach(), g["logits"]), "loss": abs(float(loss.detach()) - float(g["loss"]))}
    pk = [str(k) for k in g["param_keys"]]
    res["grad_norm"] = max(abs(float(sdr[k].grad.double().norm()) - g["grad_norms"][i]) / max(g["grad_norms"][i], 1e-4 * max(g["grad_norms"]))
                           for i, k in enumerate(pk))
    full = {k[2:]: g[k] for k in g.files if k.startswith("g:")}
    res["grad_full"] = max(rel_err(sdr[k].grad, v) for k, v in full.items() if abs(v).max() > 1e-6 * max(abs(u).max() for u in full.values()))
    assert res["logits"] < 1e-5 and res["loss"] < 1e-5 and res["grad_norm"] < 1e-3 and res["grad_full"] < 1e-3, res
    return res


def run(dev, mode):
    """The engine in `mode` (training mode, the golden's dropout masks injected) against the golden and, element by element for
    every parameter gradient, against the oracle."""
    import cbim_amd
    from cbim_amd import functional as Fn
    from cbim_amd.model.dim3 import vnet as vmod
    from oracle import loss_ref, vnet_ref
    g = load_golden("vnet_b8")
    net, sd = _golden_state_dict(g)
    x, lab, w = torch.from_numpy(g["x"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    lo = vnet_ref.vnet_forward(sdr, x, SCALE, masks=_masks(g))
    loss_ref.ce_dice_loss(lo, lab, w).backward()
    masks = _masks(g)
    orig = vmod.dropout3d_mask

    def injected(n, c, p, training, device):
        assert training and p == 0.5
        m = masks.pop(0)
        assert tuple(m.shape) == (n, c), (tuple(m.shape), n, c)
        return m.to(device)

    cbim_amd.set_compute_dtype(mode)
    vmod.dropout3d_mask = injected
    try:
        net = net.to(dev).train()
        logits = net(x.to(dev))
        both = Fn.DiceCEFn.apply(logits, lab.to(dev), w.to(dev))
        both[2].backward()
    finally:
        vmod.dropout3d_mask = orig
        cbim_amd.set_compute_dtype(None)
    assert not masks, "not every recorded dropout mask was consumed"
    pk = [str(k) for k in g["param_keys"]]
    params = dict(net.named_parameters())
    worst, cos_min, n_ok, n_t, worst_k = grad_compare({k: params[k].grad for k in pk}, {k: sdr[k].grad for k in pk})
    sdn = net.state_dict()
    res = {
        "logits_err": rel_err(logits.detach().cpu(), g["logits"]), "loss_err": abs(float(both[2]) - float(g["loss"])),
        "ce_err": abs(float(both[0]) - float(g["ce"])), "dice_err": abs(float(both[1]) - float(g["dice"])),
        "argmax_mismatch": int((logits.argmax(1).cpu() != torch.from_numpy(g["logits"]).argmax(1)).sum()),
        "grad_norm_err": max(abs(float(params[k].grad.double().norm()) - g["grad_norms"][i]) / max(g["grad_norms"][i], 1e-4 * max(g["grad_norms"]))
                             for i, k in enumerate(pk)),
        "grad_rel_worst": worst, "grad_cos_min": cos_min, "grad_rel_worst_tensor": worst_k, "grad_tensors_within_1e3": n_ok, "grad_tensors": n_t,
        "running_mean_err": max(rel_err(sdn["in_tr.bn1.running_mean"].cpu(), g["rm:in_tr.bn1"]),
                                rel_err(sdn["up_tr64.ops.0.bn1.running_mean"].cpu(), g["rm:up_tr64.ops.0.bn1"])),
        "running_var_err": max(rel_err(sdn["in_tr.bn1.running_var"].cpu(), g["rv:in_tr.bn1"]),
                               rel_err(sdn["up_tr64.ops.0.bn1.running_var"].cpu(), g["rv:up_tr64.ops.0.bn1"])),
    }
    return res


def assert_fp32(dev):
    r = run(dev, "fp32")
    assert r["logits_err"] < 1e-3 and r["argmax_mismatch"] == 0, r
    assert r["loss_err"] < 1e-4 and r["ce_err"] < 1e-4 and r["dice_err"] < 1e-4, r
    assert r["grad_norm_err"] < 1e-2 and r["grad_rel_worst"] < 5e-2 and r["grad_cos_min"] > 0.9999, r
    assert r["running_mean_err"] < 1e-4 and r["running_var_err"] < 1e-4, r
    return r
