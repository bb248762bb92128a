"""Golden fixture for VNet.  Run in the build container only (needs /root/reference):  python tests/golden/make_golden_vnet.py

The reference's model/dim3/vnet.py is executed UNMODIFIED in TRAINING mode (Dropout3d active, ContBatchNorm3d with batch
statistics).  The dropout masks are made reproducible on any device by replacing torch.nn.functional.dropout3d for the
duration of the run with `x * mask_k`, mask_k = Bernoulli(0.5) / 0.5 per (sample, channel) drawn from a seeded CPU generator in
call order — exactly what F.dropout3d computes, with the random stream under our control; the masks are stored.

  vnet_b8   baseChans 8, in_chan 1, 4 classes, scale [[1,2,2],[2,2,2],[2,2,2],[2,2,2]] (config/acdc/vnet_3d.yaml's structure at
            half its width), 2 x 1 x 16 x 32 x 32: seeded weights (checksum), logits, CE / Dice, per-parameter gradient norms and
            sums, every gradient tensor of <= 20000 elements in full, the running statistics after the step.
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402

CASES = {"vnet_b8": (1, 8, 4, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], (16, 32, 32), 2, 5051)}


def main():
    _, DiceLoss = mg.import_reference()
    VNet = importlib.import_module("model.dim3.vnet").VNet
    from oracle.unet_ref import state_dict_checksum
    torch.set_num_threads(8)
    for name, (in_ch, base, classes, scale, shape, batch, seed) in CASES.items():
        torch.manual_seed(seed)
        net = VNet(in_ch, classes, scale=scale, baseChans=base)       # model/utils.py:74 call shape
        net.train()
        gen = torch.Generator().manual_seed(seed + 1)
        x = torch.randn((batch, in_ch) + shape, generator=gen).clamp_(-7.4, 2.2)
        lab = mg.make_labels(classes, shape, batch, gen)
        weight = torch.ones(classes)
        weight[0] = 0.5
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        mgen = torch.Generator().manual_seed(seed + 2)
        masks = []
        orig = F.dropout3d

        def fake_dropout3d(inp, p=0.5, training=True, inplace=False):
            assert training and p == 0.5
            m = (torch.rand(inp.shape[0], inp.shape[1], generator=mgen) >= p).float() / (1.0 - p)
            masks.append(m)
            return inp * m.view(m.shape[0], m.shape[1], 1, 1, 1)

        F.dropout3d = fake_dropout3d
        try:
            logits = net(x)
        finally:
            F.dropout3d = orig
        ce = torch.nn.CrossEntropyLoss(weight=weight)(logits, lab.squeeze(1))
        dl = DiceLoss()(logits, lab)
        loss = ce + dl
        loss.backward()
        grads = {k: p.grad for k, p in net.named_parameters()}
        pkeys = [k for k, _ in net.named_parameters()]
        sd1 = net.state_dict()
        out = {
            "x": x.numpy(), "label": lab.numpy().astype(np.int64), "weight": weight.numpy(), "logits": logits.detach().numpy(),
            "ce": np.float64(ce.item()), "dice": np.float64(dl.item()), "loss": np.float64(loss.item()),
            "n_params": np.int64(sum(p.numel() for p in net.parameters())), "n_tensors": np.int64(len(sd0)),
            "keys": np.array(list(sd0.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd0.values()]),
            "param_keys": np.array(pkeys), "n_masks": np.int64(len(masks)),
            "grad_norms": np.array([float(grads[k].double().norm()) for k in pkeys]),
            "grad_sums": np.array([float(grads[k].double().sum()) for k in pkeys]),
            "sd_checksum": np.float64(state_dict_checksum({k: sd0[k] for k in pkeys})), "seed": np.int64(seed),
            "rm:in_tr.bn1": sd1["in_tr.bn1.running_mean"].numpy(), "rv:in_tr.bn1": sd1["in_tr.bn1.running_var"].numpy(),
            "rm:up_tr64.ops.0.bn1": sd1["up_tr64.ops.0.bn1.running_mean"].numpy(),
            "rv:up_tr64.ops.0.bn1": sd1["up_tr64.ops.0.bn1.running_var"].numpy(),
        }
        for i, m in enumerate(masks):
            out[f"mask{i}"] = m.numpy()
        for k in pkeys:
            if grads[k].numel() <= 20000:
                out["g:" + k] = grads[k].numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "logits", tuple(logits.shape), "loss", float(loss), "params", int(out["n_params"]), "tensors", int(out["n_tensors"]),
              "masks", len(masks), [tuple(m.shape) for m in masks], "size", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
