"""Golden fixture for AttentionUNet (model/dim3/attention_unet.py of the REAL reference, CPU fp32).
    python tests/golden/make_golden_attunet.py
Anisotropic first level, base_chan 8, 16x32x32, seeded weights; conv_ch (declared, unused by the reference) gets no gradient."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402

SEED, SHAPE, CLASSES, BASE = 7071, (16, 32, 32), 4, 8
SCALE = [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
KS = [[1, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]]


def main():
    _, DiceLoss = mg.import_reference()
    AttUNet = importlib.import_module("model.dim3.attention_unet").AttentionUNet
    from oracle.unet_ref import state_dict_checksum
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    net = AttUNet(1, BASE, scale=SCALE, kernel_size=KS, num_classes=CLASSES, block="BasicBlock", norm="in")
    net.train()
    gen = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn((1, 1) + SHAPE, generator=gen).clamp_(-7.4, 2.2)
    lab = mg.make_labels(CLASSES, SHAPE, 1, gen)
    weight = torch.ones(CLASSES)
    weight[0] = 0.5
    logits = net(x)
    ce = torch.nn.CrossEntropyLoss(weight=weight)(logits, lab.squeeze(1))
    dl = DiceLoss()(logits, lab)
    (ce + dl).backward()
    sd = net.state_dict()
    grads = {k: p.grad for k, p in net.named_parameters()}
    out = dict(x=x.numpy(), label=lab.numpy().astype(np.int64), weight=weight.numpy(), logits=logits.detach().numpy(),
               ce=np.float64(ce.item()), dice=np.float64(dl.item()), keys=np.array(list(sd.keys())),
               shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
               n_params=np.int64(sum(p.numel() for p in net.parameters())),
               grad_norms=np.array([float(grads[k].double().norm()) if grads[k] is not None else -1.0 for k in sd.keys()]),
               sd_checksum=np.float64(state_dict_checksum(sd)), seed=np.int64(SEED))
    out["g:inc.conv1.weight"] = grads["inc.conv1.weight"].numpy()
    out["g:outc.weight"] = grads["outc.weight"].numpy()
    path = os.path.join(HERE, "attunet_b8.npz")
    np.savez_compressed(path, **out)
    print("logits", tuple(logits.shape), float(ce + dl), int(out["n_params"]), os.path.getsize(path) // 1024, "KB")


NORM_CASES = {
    # name: (block, norm, batch, seed) — round 5: `norm: bn | ln` reach the BLOCKS only; the gate keeps nn.InstanceNorm3d (attention_unet_utils.py:10-21)
    "attunet_bn_b8": ("SingleConv", "bn", 2, 7081),
    "attunet_ln_b8": ("BasicBlock", "ln", 1, 7082),
}


def main_norms(only=()):
    _, DiceLoss = mg.import_reference()
    AttUNet = importlib.import_module("model.dim3.attention_unet").AttentionUNet
    from oracle.unet_ref import state_dict_checksum
    torch.set_num_threads(8)
    for name, (block, norm, batch, seed) in NORM_CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(seed)
        net = AttUNet(1, BASE, scale=SCALE, kernel_size=KS, num_classes=CLASSES, block=block, norm=norm)
        gen = torch.Generator().manual_seed(seed + 1)
        affine = {}
        with torch.no_grad():
            for k, p in net.named_parameters():
                if k.endswith("norm.weight"):
                    p.copy_(1.0 + 0.5 * torch.randn(p.shape, generator=gen))
                    affine[k] = p.detach().clone()
                elif k.endswith("norm.bias"):
                    p.copy_(0.3 * torch.randn(p.shape, generator=gen))
                    affine[k] = p.detach().clone()
        sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net.train()
        x = torch.randn((batch, 1) + SHAPE, generator=gen).clamp_(-7.4, 2.2)
        lab = mg.make_labels(CLASSES, SHAPE, batch, gen)
        weight = torch.ones(CLASSES)
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
        out = dict(x=x.numpy(), label=lab.numpy().astype(np.int64), weight=weight.numpy(), logits=logits.detach().numpy(),
                   logits_eval=logits_eval.numpy(), ce=np.float64(ce.item()), dice=np.float64(dl.item()),
                   keys=np.array(list(sd0.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd0.values()]),
                   param_keys=np.array([k for k, _ in net.named_parameters()]),
                   grad_norms=np.array([float(grads[k].double().norm()) if grads[k] is not None else -1.0 for k in grads]),
                   sd_checksum=np.float64(state_dict_checksum({k: v for k, v in sd0.items() if v.is_floating_point()})),
                   seed=np.int64(seed))
        for k, v in affine.items():
            out["p:" + k] = v.numpy()
        for k, g in grads.items():           # full gradients of the first layers, one gate and the head
            if g is not None and (k.startswith("inc.") or k.startswith("up4.attn.") or k.startswith("outc.")):
                out["g:" + k] = g.numpy()
        for k, v in sd1.items():
            if k.startswith("inc.") and ("running_" in k or "num_batches" in k):
                out["r:" + k] = v.numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "logits", tuple(logits.shape), float(ce + dl), len(sd0), "tensors", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "norms":
        main_norms(sys.argv[2:])
    else:
        main()
