"""Helper of tests/test_reference_trainer_seam.py (run as a subprocess so that the reference's top-level package names
`model`, `training`, `utils`, ... never enter the test process).

Imports the REFERENCE'S OWN trainer (/root/reference/train.py, unmodified) with exactly the one-line seam of
INTEGRATION.md §1 applied — `model.utils.get_model` resolves to `cbim_amd.model.utils.get_model` — and drives
`init_network` (train.py:277-300) and `train_epoch` (train.py:138-233) for a few iterations: the reference's loop, its
`nn.CrossEntropyLoss` + its own `training.losses.DiceLoss` on the engine's logits, its `training.utils.get_optimizer`
(torch AdamW) on the engine's parameters, its `update_ema_variables`.  Test infrastructure only.
"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    sys.path.insert(0, ROOT)
    import torch
    import cbim_amd
    from cbim_amd.model import utils as amd_model_utils

    # ---- the seam: `from model.utils import get_model` (train.py:11) gets the engine's get_model ---------------
    sys.path.insert(0, REF)
    pkg = types.ModuleType("model")
    pkg.__path__ = [os.path.join(REF, "model")]
    sys.modules["model"] = pkg
    shim = types.ModuleType("model.utils")
    shim.get_model = amd_model_utils.get_model
    sys.modules["model.utils"] = shim
    # ---- things the trainer imports that this image does not have / that need a dataset on disk ------------------
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            self.scalars = []

        def add_scalar(self, tag, value, step):
            self.scalars.append((tag, float(value), step))

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    for missing in ("SimpleITK", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    ds = types.ModuleType("training.dataset.utils")
    ds.get_dataset = lambda args, mode, **kw: None          # train_epoch gets its loader from this script
    import training                                          # the reference's package
    dpk = types.ModuleType("training.dataset")
    dpk.__path__ = []
    sys.modules["training.dataset"] = dpk
    sys.modules["training.dataset.utils"] = ds

    import train as ref_train                                # /root/reference/train.py, unmodified
    import yaml
    import argparse
    from training.losses import DiceLoss as RefDiceLoss      # the reference's own loss on the engine's logits
    from training.utils import get_optimizer as ref_get_optimizer

    with open(os.path.join(REF, "config/amos_ct/resunet_3d.yaml")) as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    args = argparse.Namespace(**cfg)
    args.dimension, args.model, args.dataset = "3d", "resunet", "amos_ct"
    args.pretrain = args.amp = args.resume = args.torch_compile = False
    args.base_chan = 4                                       # the executor is slow: shrink widths / crop, keep the structure
    args.training_size = [16, 16, 16]
    args.aug_device = "gpu"                                  # tensors are handed over where the engine lives (train.py:153)
    args.iter_per_epoch, args.print_freq = 1, 1
    on_gpu = torch.cuda.is_available() and cbim_amd._lib.backend() != "emu"
    dev = torch.device("cuda", 0) if on_gpu else torch.device("cpu")

    torch.manual_seed(2023)
    net, ema_net = ref_train.init_network(args)              # train.py:277-300 -> get_model twice through the seam
    assert type(net).__module__.startswith("cbim_amd."), type(net).__module__
    net, ema_net = net.to(dev), ema_net.to(dev)
    w0 = [p.detach().clone() for p in net.parameters()]
    e0 = [p.detach().clone() for p in ema_net.parameters()]
    optimizer = ref_get_optimizer(args, net)                 # training/utils.py:8-14 (torch AdamW, eps 1e-5)
    criterion = torch.nn.CrossEntropyLoss(weight=torch.tensor(args.weight).to(dev))
    criterion_dl = RefDiceLoss()
    g = torch.Generator().manual_seed(7)
    loader = [(torch.randn(1, 1, 16, 16, 16, generator=g).to(dev),
               torch.randint(0, args.classes, (1, 1, 16, 16, 16), generator=g).to(torch.int8).to(dev)) for _ in range(3)]
    writer = SummaryWriter()
    calls = []
    net.register_forward_hook(lambda m, i, o: calls.append(1))
    ref_train.train_epoch(loader, net, ema_net, optimizer, 0, writer, criterion, criterion_dl, None, args)
    moved = sum(float((p.detach() - w).abs().sum()) for p, w in zip(net.parameters(), w0))
    ema_moved = sum(float((p.detach() - w).abs().sum()) for p, w in zip(ema_net.parameters(), e0))
    losses = [v for t, v, _ in writer.scalars if t == "Train/Loss"]
    print("SEAM " + json.dumps({"iters": len(calls), "losses": losses, "moved": moved, "ema_moved": ema_moved,
                                "net": type(net).__module__ + "." + type(net).__name__,
                                "finite": all(bool(torch.isfinite(p).all()) for p in net.parameters())}))


if __name__ == "__main__":
    main()
