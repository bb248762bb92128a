"""Golden fixture for sliding-window inference + evaluation Dice, produced by EXECUTING THE REAL REFERENCE
(inference/inference3d.py, metric/utils.py) on CPU.   python tests/golden/make_golden_infer.py

A seeded ResUNet (base 8, 3 classes; weights reproducible from the seed through oracle.unet_ref) is run over a
40x48x40 volume with 32^3 half-overlapping windows; stored: the averaged probabilities, the argmax map,
and calculate_dice / calculate_dice_split of that map against blocky labels (block_size 30000 so that several
blocks and a ragged tail occur)."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402

SEED, SHAPE, WINDOW, CLASSES, BASE, BLOCK = 5051, (40, 48, 40), [32, 32, 32], 3, 8, 30000


def main():
    UNet, _ = mg.import_reference()
    inf = importlib.import_module("inference.inference3d")
    # metric/utils.py imports the surface-distance package at module level; only the dice functions are needed
    import types
    pkg = types.ModuleType("metric"); pkg.__path__ = ["/root/reference/metric"]; sys.modules["metric"] = pkg
    sys.modules["metric.metrics"] = types.ModuleType("metric.metrics")
    mu = importlib.import_module("metric.utils")
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    net = UNet(1, BASE, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=CLASSES, block="BasicBlock", norm="in")
    gen = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn((1, 1) + SHAPE, generator=gen).clamp_(-7.4, 2.2)
    lab = mg.make_labels(CLASSES, SHAPE, 1, gen)
    args = argparse.Namespace(window_size=WINDOW, classes=CLASSES, dimension="3d", sliding_window=True)
    prob = inf.inference_sliding_window(net, x, args)
    whole = inf.inference_whole_image(net, x[:, :, :32, :32, :32].contiguous())
    _, label_pred = torch.max(prob, dim=1)
    d1, i1, s1 = mu.calculate_dice(label_pred.view(-1, 1), lab.view(-1, 1), CLASSES)
    d2, i2, s2 = mu.calculate_dice_split(label_pred.view(-1, 1), lab.view(-1, 1), CLASSES, block_size=BLOCK)
    from oracle.unet_ref import state_dict_checksum
    out = dict(x=x.numpy(), label=lab.numpy().astype(np.int64), prob=prob.numpy(), whole=whole.numpy(),
               label_pred=label_pred.numpy().astype(np.int64), dice=d1.numpy(), inter=i1.numpy(), summ=s1.numpy(),
               dice_split=d2.numpy(), inter_split=i2.numpy(), summ_split=s2.numpy(),
               sd_checksum=np.float64(state_dict_checksum(net.state_dict())), seed=np.int64(SEED))
    path = os.path.join(HERE, "infer_resunet_b8.npz")
    np.savez_compressed(path, **out)
    print("prob", tuple(prob.shape), "dice", d1.numpy(), d2.numpy(), "size", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
