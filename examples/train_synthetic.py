"""The reference's training loop (/root/reference/train.py:84-232 with validation.py:16-60) on the MI355X engine, every
step through this package's drop-in surfaces — on synthetic volumes, because the image has no datasets:

    get_model(args)                                   model/utils.py:6
    ResidentVolumeDataset + DevicePrefetcher          dataset_amos_ct.py:105-165 recipe, on the device
    DiceCELoss (= CrossEntropyLoss(weight) + DiceLoss)  train.py:80-81,206-212
    get_optimizer / exp_lr_scheduler_with_warmup / update_ema_variables   training/utils.py, train.py:94,216-218
    inference_sliding_window + calculate_dice         training/validation.py:38-60 on the EMA net (train.py:101)

    python examples/train_synthetic.py --config amos_ct/resunet_3d.yaml --iters 50          # on an MI355X

The YAML keys come from tests/golden/shipped_configs.json (the values of the reference's own config files); --base /
--size shrink the model and the crop for a quick run.  tests/test_train_loop_example.py runs it on the host-side executor.
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_volume(classes, size, gen, device):
    """Blocky label map (every class present) and an image whose intensity depends on the label, AMOS value range."""
    coarse = [max(s // 8, 1) for s in size]
    lab = torch.randint(0, classes, (1, 1, *coarse), generator=gen).float()
    lab = torch.nn.functional.interpolate(lab, size=tuple(size), mode="nearest").to(torch.int8)[0, 0]
    img = (lab.float() / max(classes - 1, 1) * 4.0 - 3.0 + 0.3 * torch.randn(tuple(size), generator=gen)).clamp_(-7.4, 2.2)
    return img.to(device), lab.to(device)


def run(config="amos_ct/resunet_3d.yaml", iters=20, epochs=2, base=None, size=None, device=None, dtype="bf16", seed=2023,
        n_volumes=3, verbose=True, sliding_window=True):
    import cbim_amd
    from cbim_amd.inference.utils import get_inference
    from cbim_amd.metric.utils import calculate_dice
    from cbim_amd.model.utils import get_model
    from cbim_amd.training.dataset.resident import DevicePrefetcher, ResidentVolumeDataset
    from cbim_amd.training.losses import DiceCELoss
    from cbim_amd.training.utils import exp_lr_scheduler_with_warmup, get_optimizer, update_ema_variables

    with open(os.path.join(ROOT, "tests", "golden", "shipped_configs.json")) as f:
        cfg = dict(json.load(f)[config]["args"])
    if base:
        cfg["base_chan"] = base
    if size:
        cfg["training_size"] = list(size)
        cfg["window_size"] = list(size)
    # the optimisation keys every shipped 3D yaml carries (e.g. config/amos_ct/resunet_3d.yaml:26-47)
    cfg.update(optimizer="adamw", base_lr=6e-4, betas=[0.9, 0.999], weight_decay=0.05, ema_alpha=0.99, epochs=epochs,
               affine_pad_size=[max(s // 4, 2) for s in cfg["training_size"]], scale=[0.3] * 3, rotate=[30] * 3,
               translate=[0] * 3, sliding_window=sliding_window)
    args = argparse.Namespace(**cfg)
    device = device or torch.device("cuda", 0)
    cbim_amd.set_compute_dtype(dtype)
    try:
        torch.manual_seed(seed)
        np.random.seed(seed)
        net = get_model(args).to(device)
        ema_net = copy.deepcopy(net)                                     # train.py:281-284
        for p in ema_net.parameters():
            p.requires_grad_(False)
        gen = torch.Generator().manual_seed(seed)
        vol_size = [t + p for t, p in zip(args.training_size, args.affine_pad_size)]
        vols = [synthetic_volume(args.classes, vol_size, gen, device) for _ in range(n_volumes)]
        feeder = DevicePrefetcher(ResidentVolumeDataset([v[0] for v in vols], [v[1] for v in vols], args))
        weight = torch.tensor(cfg.get("weight", [1.0] * args.classes), dtype=torch.float32)
        criterion = DiceCELoss(weight).to(device)
        optimizer = get_optimizer(args, net)
        aux_w = cfg.get("aux_weight", [0.5, 0.5])
        losses, step = [], 0
        for epoch in range(args.epochs):
            lr = exp_lr_scheduler_with_warmup(optimizer, init_lr=args.base_lr, epoch=epoch, warmup_epoch=5, max_epoch=args.epochs)
            net.train()
            for _ in range(iters):
                img, label = feeder.next()
                result = net(img)
                if isinstance(result, (tuple, list)):                    # train.py:206-210
                    loss = sum(aw * criterion(r, label) for aw, r in zip(aux_w, result))
                else:
                    loss = criterion(result, label)
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                update_ema_variables(net, ema_net, args.ema_alpha, step)  # train.py:218
                step += 1
                losses.append(float(loss.detach()))
            if verbose:
                print(f"epoch {epoch}: lr {lr:.3e}  loss {np.mean(losses[-iters:]):.4f}")
        # validation on the EMA net (train.py:101, validation.py:38-60): sliding window over a whole volume, hard Dice
        ema_net.eval()
        inference = get_inference(args)
        img, lab = vols[0]
        with torch.no_grad():
            prob = inference(ema_net, img[None, None], args)
            pred = prob.argmax(1)
        dice, _, _ = calculate_dice(pred.reshape(-1, 1), lab.long().reshape(-1, 1), args.classes)   # validation.py:52-60
        if verbose:
            print("EMA-net Dice per class:", [round(float(d), 3) for d in dice])
        return {"losses": losses, "dice": [float(d) for d in dice], "lr": lr, "steps": step}
    finally:
        cbim_amd.set_compute_dtype(None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="amos_ct/resunet_3d.yaml")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--base", type=int, default=None)
    ap.add_argument("--size", type=int, nargs=3, default=None)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    run(a.config, a.iters, a.epochs, a.base, a.size, dtype=a.dtype)
