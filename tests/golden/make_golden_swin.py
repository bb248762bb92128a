"""Golden fixtures for SwinUNETR.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_swin.py

The reference's swin_unetr.py is executed UNMODIFIED; the five `monai==1.1.0` blocks it imports are not
installed here, so they come from the torch-only stand-in in tests/golden/monai_standin (written from MONAI
1.1.0's published behaviour).  Hence: the Swin transformer part of these fixtures is the real reference, the
MONAI conv blocks are "parity unpinned" (SURVEY.md §8c).  Cases
  swin_tiny        feature_size 24, in_chan 4, 3 classes, 64x32x32 (anisotropic windows, one shifted stage with a
                   window smaller than 7 in two dims), seeded weights, every gradient tensor of <= 20000 elements in full, norms/sums of all, the 5 hidden states
  swin_c1_tiny     in_chan 1 / 14 classes as in config/bcv/swin_unetr_3d.yaml, feature_size 24, 64x32x32
  swin_brats_64    feature_size 48, in_chan 4, 4 classes (BASELINE config 5 shape) at 64^3; seeded weights,
                   strided logits, losses, per-parameter gradient norms
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "monai_standin"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402

CASES = {
    # name: (img_size, in_chan, classes, feature_size, batch, seed, full)
    "swin_tiny": ((64, 32, 32), 4, 3, 24, 1, 4041, True),
    "swin_brats_64": ((64, 64, 64), 4, 4, 48, 1, 4042, False),
    # the shipped configs (config/{bcv,kits,lits}/swin_unetr_3d.yaml) are single-modality: in_chan 1, 14 classes (bcv)
    "swin_c1_tiny": ((64, 32, 32), 1, 14, 24, 1, 4043, True),
}


def main():
    _, DiceLoss = mg.import_reference()
    SwinUNETR = importlib.import_module("model.dim3.swin_unetr").SwinUNETR
    from oracle.unet_ref import state_dict_checksum
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (shape, in_ch, classes, feat, batch, seed, full) in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(seed)
        net = SwinUNETR(shape, in_ch, classes, feature_size=feat)     # model/utils.py:113 call shape
        net.train()
        gen = torch.Generator().manual_seed(seed + 1)
        x = torch.randn((batch, in_ch) + shape, generator=gen).clamp_(-7.4, 2.2)
        lab = mg.make_labels(classes, shape, batch, gen)
        weight = torch.ones(classes)
        weight[0] = 0.5
        hidden = net.swinViT(x, net.normalize)
        logits = net(x)
        ce = torch.nn.CrossEntropyLoss(weight=weight)(logits, lab.squeeze(1))
        dl = DiceLoss()(logits, lab)
        loss = ce + dl
        loss.backward()
        sd = {k: v for k, v in net.state_dict().items()}
        grads = {k: p.grad for k, p in net.named_parameters()}
        pkeys = [k for k, _ in net.named_parameters()]
        st = (2 if classes > 8 else 1) if full else 4    # keep the fixture small
        out = {
            "x": x.numpy(), "label": lab.numpy().astype(np.int64), "weight": weight.numpy(),
            "logits": logits.detach().numpy()[..., ::st, ::st, ::st], "stride": np.int64(st),
            "ce": np.float64(ce.item()), "dice": np.float64(dl.item()), "loss": np.float64(loss.item()),
            "n_params": np.int64(sum(p.numel() for p in net.parameters())), "n_tensors": np.int64(len(sd)),
            "n_buffers": np.int64(len(list(net.buffers()))),
            "keys": np.array(list(sd.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd.values()]),
            "param_keys": np.array(pkeys),
            "grad_norms": np.array([float(grads[k].double().norm()) for k in pkeys]),
            "grad_sums": np.array([float(grads[k].double().sum()) for k in pkeys]),
            "sd_checksum": np.float64(state_dict_checksum({k: sd[k] for k in pkeys})), "seed": np.int64(seed),
            "g:out.conv.conv.weight": grads["out.conv.conv.weight"].numpy(),
            "g:encoder1.layer.conv1.conv.weight": grads["encoder1.layer.conv1.conv.weight"].numpy(),
            "g:swinViT.patch_embed.proj.weight": grads["swinViT.patch_embed.proj.weight"].numpy(),
        }
        if full:   # weights come from the seed (checksum-verified); keep every small gradient tensor in full
            for k in pkeys:
                if grads[k].numel() <= 20000:
                    out["g:" + k] = grads[k].numpy()
            for i, hsv in enumerate(hidden):
                out[f"hidden{i}"] = hsv.detach().numpy().astype(np.float32)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "logits", tuple(logits.shape), "loss", float(loss), "params", int(out["n_params"]), "tensors",
              int(out["n_tensors"]), "size", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
