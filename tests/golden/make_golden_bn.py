"""Golden fixtures for the non-default constructor branches `norm: bn` and `pool=False` — the `norm: bn` constructor branch (nn.BatchNorm3d in every ConvNormAct, /root/reference/model/dim3/
utils.py:15-21, conv_layers.py:40-43), produced by EXECUTING THE REAL REFERENCE on CPU (build container only):

    python tests/golden/make_golden_bn.py

The reference UNet is built with norm='bn' under a fixed seed, its BatchNorm affine parameters are perturbed (defaults are
gamma = 1, beta = 0: the affine path would not be exercised), and one TRAINING step's forward + CE + Dice + backward is recorded
(batch statistics, running-statistics update), then the logits of the same input in EVAL mode (running statistics).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests.golden.make_golden import import_reference, make_labels  # noqa: E402

CASES = {
    # name: (in_ch, base_ch, classes, scale, kernel_size, block, spatial, batch, seed, norm, pool)
    "resunet_bn_b8": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 32), 2, 3031, "bn", True),
    "unet_single_bn_b8": (2, 8, 3, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "SingleConv", (8, 32, 32), 2, 3032, "bn", True),
    # down_block(pool=False): the first block of every level strides (unet_utils.py:36-39) — InstanceNorm ResUNet, an odd extent
    # along W at the second level (36 -> 18 -> 9 -> 5 -> 3), and a BatchNorm SingleConv UNet with an anisotropic first stride
    "resunet_nopool_b8": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 36), 1, 3033, "in", False),
    "unet_single_nopool_bn": (1, 8, 3, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "SingleConv", (8, 32, 32), 2, 3034, "bn", False),
    "resunet_bottleneck_nopool_b16": (1, 16, 3, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "Bottleneck", (32, 32, 32), 1, 3035, "in", False),
    # `norm: ln`: the channels-first LayerNorm (trans_layers.py:120-149) in every ConvNormAct
    "resunet_ln_b8": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 32), 1, 3036, "ln", True),
    "unet_single_ln_b8": (2, 8, 3, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "SingleConv", (8, 32, 32), 2, 3037, "ln", True),
}


def main():
    UNet, DiceLoss = import_reference()
    from oracle.unet_ref import state_dict_checksum
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (in_ch, base, classes, scale, ks, block, shape, batch, seed, norm, pool) in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(seed)
        net = UNet(in_ch, base, scale=scale, kernel_size=ks, num_classes=classes, block=block, norm=norm, pool=pool)
        gen = torch.Generator().manual_seed(seed + 1)
        affine = {}
        with torch.no_grad():
            for k, p in net.named_parameters():
                if k.endswith("norm.weight"):
                    p.copy_(1.0 + 0.5 * torch.randn(p.shape, generator=gen))
                    p[0] = 1e-3                       # a nearly dead channel and a negative gamma
                    if p.numel() > 1:
                        p[1] = -0.6
                    affine[k] = p.detach().clone()
                elif k.endswith("norm.bias"):
                    p.copy_(0.3 * torch.randn(p.shape, generator=gen))
                    affine[k] = p.detach().clone()
        sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net.train()
        x = torch.randn((batch, in_ch) + shape, generator=gen).clamp_(-7.4, 2.2)
        lab = make_labels(classes, shape, batch, gen)
        weight = torch.ones(classes)
        weight[0] = 0.5
        logits = net(x)
        ce = torch.nn.CrossEntropyLoss(weight=weight)(logits, lab.squeeze(1))
        dl = DiceLoss()(logits, lab)
        (ce + dl).backward()
        grads = {k: p.grad for k, p in net.named_parameters()}
        sd1 = net.state_dict()
        net.eval()
        with torch.no_grad():
            logits_eval = net(x)
        out = {
            "x": x.numpy(), "label": lab.numpy().astype(np.int64), "weight": weight.numpy(),
            "logits": logits.detach().numpy(), "logits_eval": logits_eval.numpy(), "ce": np.float64(ce.item()), "dice": np.float64(dl.item()),
            "keys": np.array(list(sd0.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd0.values()]),
            "param_keys": np.array([k for k, _ in net.named_parameters()]),
            "sd_checksum": np.float64(state_dict_checksum({k: v for k, v in sd0.items() if v.is_floating_point()})),
            "grad_norms": np.array([float(grads[k].double().norm()) for k in grads]),
        }
        for k, v in affine.items():
            out["p:" + k] = v.numpy()
        for k, g in grads.items():                        # full gradients of every BatchNorm parameter, the stem and the head
            if "norm." in k or k in ("inc.conv1.weight", "outc.weight", "outc.bias") or ".conv.1." in k and k.startswith("down1"):
                out["g:" + k] = g.numpy()
        for k, v in sd1.items():                          # running statistics after the training step
            if "running_" in k or "num_batches" in k:
                out["r:" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "logits", tuple(logits.shape), "loss", float(ce + dl), "tensors", len(sd0), "size",
              os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
