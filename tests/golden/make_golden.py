"""Generate the golden fixtures by EXECUTING THE REAL REFERENCE on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's own modules through the import shim of SURVEY.md §8c
(skipping model/dim3/__init__.py, which pulls monai/timm/mmcv), builds the model with the
reference constructor under a fixed torch seed, runs forward + CE+Dice loss + backward in
fp32, and writes small .npz fixtures next to this file.  The fixtures pin
  (1) the oracle restatement (tests/test_oracle.py, runs everywhere), and
  (2) the HIP path on the GPU box (tests/test_gpu_parity.py), where /root/reference is absent.
No reference source is copied; only tensors it produced.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.path.insert(0, REF)
    for name, path in [("model", f"{REF}/model"), ("model.dim3", f"{REF}/model/dim3")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg
    for missing in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    UNet = importlib.import_module("model.dim3.unet").UNet
    from training.losses import DiceLoss
    return UNet, DiceLoss


CASES = {
    # name: (in_ch, base_ch, classes, scale, kernel_size, block, spatial, batch, seed, full_sd)
    "resunet_b2_32": (1, 2, 3, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 32), 2, 2023, True),
    "resunet_b8_32": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 32), 1, 2024, False),
    "resunet_b8_aniso": (2, 8, 5, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "BasicBlock",
                         (8, 48, 32), 1, 2025, False),
    # ACDC yaml kernel/scale (config/acdc/unet_3d.yaml:11-14): even kernel [2,3,3] grows D by 1 per conv
    "unet_single_acdc": (1, 8, 4, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]],
                         [[1, 3, 3], [2, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]], "SingleConv",
                         (16, 32, 32), 1, 2026, False),
    # UNet(block='Bottleneck') (conv_layers.py:96-125): 1x1 -> 3^3 -> 1x1 with expansion 2, full-size conv shortcuts
    "resunet_bottleneck_b16": (1, 16, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "Bottleneck", (32, 32, 32), 1, 2027, False),
}


def make_labels(classes, shape, batch, gen):
    # blocky labels so every class is present
    coarse = torch.randint(0, classes, (batch, 1) + tuple(max(1, s // 4) for s in shape), generator=gen)
    lab = torch.nn.functional.interpolate(coarse.float(), size=shape, mode="nearest").long()
    return lab


def main():
    UNet, DiceLoss = import_reference()
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (in_ch, base, classes, scale, ks, block, shape, batch, seed, full_sd) in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(seed)
        net = UNet(in_ch, base, scale=scale, kernel_size=ks, num_classes=classes, block=block, norm="in")
        net.train()
        gen = torch.Generator().manual_seed(seed + 1)
        x = torch.randn((batch, in_ch) + shape, generator=gen).clamp_(-7.4, 2.2)
        lab = make_labels(classes, shape, batch, gen)
        weight = torch.ones(classes)
        weight[0] = 0.5
        logits = net(x)
        ce = torch.nn.CrossEntropyLoss(weight=weight)(logits, lab.squeeze(1))
        dl = DiceLoss()(logits, lab)
        loss = ce + dl
        loss.backward()
        sd = net.state_dict()
        grads = {k: p.grad for k, p in net.named_parameters()}
        out = {
            "x": x.numpy(), "label": lab.numpy().astype(np.int64), "weight": weight.numpy(),
            "logits": logits.detach().numpy(), "ce": np.float64(ce.item()), "dice": np.float64(dl.item()),
            "n_params": np.int64(sum(p.numel() for p in net.parameters())),
            "n_tensors": np.int64(len(sd)),
            "keys": np.array(list(sd.keys())),
            "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]),
            "grad_norms": np.array([float(grads[k].double().norm()) for k in sd.keys()]),
            "grad_sums": np.array([float(grads[k].double().sum()) for k in sd.keys()]),
            "g:inc.conv1.weight": grads["inc.conv1.weight"].numpy(),
            "g:outc.weight": grads["outc.weight"].numpy(),
            "g:outc.bias": grads["outc.bias"].numpy(),
        }
        sys.path.insert(0, os.path.join(HERE, "..", ".."))
        from oracle.unet_ref import state_dict_checksum
        out["sd_checksum"] = np.float64(state_dict_checksum(sd))
        if full_sd:
            for k, v in sd.items():
                out["p:" + k] = v.numpy()
                out["g:" + k] = grads[k].numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "logits", tuple(logits.shape), "loss", float(loss), "params", int(out["n_params"]),
              "size", os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KB")

    # pinned-by-construction facts: parameter counts of the shipped configs
    facts = {}
    net = UNet(1, 32, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=16, block="BasicBlock", norm="in")
    facts["resunet_amos"] = (sum(p.numel() for p in net.parameters()), len(net.state_dict()), len(list(net.buffers())))
    net = UNet(1, 32, scale=[[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]],
               kernel_size=[[1, 3, 3], [2, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]], num_classes=4, block="SingleConv", norm="in")
    facts["unet_acdc"] = (sum(p.numel() for p in net.parameters()), len(net.state_dict()), len(list(net.buffers())))
    # DiceLoss smoke values of the reference's own __main__ block (training/losses.py:100-119), seeded
    torch.manual_seed(7)
    pred = torch.randn(2, 10, 8, 16, 16)
    target = torch.zeros(2, 1, 8, 16, 16).long()
    facts_dl = float(DiceLoss()(pred, target))
    np.savez(os.path.join(HERE, "facts.npz"), resunet_amos=np.array(facts["resunet_amos"]),
             unet_acdc=np.array(facts["unet_acdc"]), dice_zero_target=np.float64(facts_dl))
    print("facts", facts, facts_dl)


if __name__ == "__main__":
    main()
