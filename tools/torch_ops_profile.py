#!/usr/bin/env python
"""Which Python lines launch the small torch kernels of a training step?  One eager step of a model under
torch.profiler with stacks; prints the ATen ops by call count with the innermost cbim_amd frame.
    python tools/torch_ops_profile.py medformer|swin_unetr|resunet"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import cbim_amd
from cbim_amd.model.dim3 import MedFormer, SwinUNETR, UNet
from cbim_amd.training.losses import DiceCELoss
from cbim_amd.training.optim import FusedAdamW

model = sys.argv[1] if len(sys.argv) > 1 else "medformer"
dev = torch.device("cuda:0")
cbim_amd.set_compute_dtype("bf16")
torch.manual_seed(0)
classes, in_ch, size = 16, 1, 128
if model == "medformer":
    net = MedFormer(1, classes, **bench.MEDFORMER_AMOS).to(dev)
elif model == "swin_unetr":
    in_ch, classes = 4, 4
    net = SwinUNETR((size,) * 3, in_ch, classes, feature_size=48).to(dev)
else:
    net = UNet(1, 32, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=classes, block="BasicBlock", norm="in").to(dev)
net.train()
crit = DiceCELoss(torch.ones(classes)).to(dev)
opt = FusedAdamW(net.parameters(), lr=6e-4)
x = torch.randn(1, in_ch, size, size, size, device=dev)
lab = torch.randint(0, classes, (1, 1, size, size, size), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    out = net(x)
    loss = sum(0.5 * crit(o, lab) for o in out) if isinstance(out, (list, tuple)) else crit(out, lab)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
NOLAUNCH = {"aten::" + n for n in ("view reshape permute transpose select slice expand as_strided t unsqueeze squeeze detach alias empty "
                                   "empty_like empty_strided _unsafe_view flatten unflatten chunk split narrow item _local_scalar_dense "
                                   "lift_fresh result_type resolve_conj resolve_neg view_as expand_as unbind split_with_sizes "
                                   "set_ _reshape_alias numpy_T is_nonzero").split()}
cnt = collections.Counter()
for ev in prof.events():
    if (ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name not in NOLAUNCH
            and not any(c.name.startswith("aten::") for c in ev.cpu_children)):
        frame = next((f for f in (ev.stack or []) if "cbim" in f and "site-packages" not in f), (ev.stack or ["?"])[0] if ev.stack else "?")
        cnt[(ev.name, frame.split("cbim-medical-image-segmentation_amd/")[-1][:90])] += 1
tot = sum(cnt.values())
print(f"{tot} leaf ATen ops (kernel-launching kinds) in one step")
for (name, frame), c in cnt.most_common(45):
    print(f"{c:5d}  {name:28s} {frame}")
