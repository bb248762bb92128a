"""VNet (config/acdc/vnet_3d.yaml: base 16, crop 16x192x192) training-step time, eager, bf16 — and the per-kernel table."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cbim_amd
from cbim_amd import ops
from cbim_amd.model.dim3 import VNet
from cbim_amd.training.losses import DiceCELoss
from cbim_amd.training.optim import FusedAdamW

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(5)
net = VNet(1, 4, scale=[[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], baseChans=16).to(dev).train()
crit = DiceCELoss(torch.tensor([0.5, 1, 1, 1.0])).to(dev)
opt = FusedAdamW(net.parameters(), lr=1e-3, weight_decay=0.05)
x = torch.randn(B, 1, 16, 192, 192, device=dev)
lab = torch.randint(0, 4, (B, 1, 16, 192, 192), device=dev)
cbim_amd.set_compute_dtype("bf16")


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(net(x), lab)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print(f"vnet acdc bf16 batch {B}: {dt*1e3:.2f} ms/step eager, {B/dt:.1f} volumes/s (16x192x192 crops)")
if len(sys.argv) > 2 and sys.argv[2] == "long":
    ls = []
    for i in range(40):
        l = step()
        if i % 4 == 0:
            ls.append(round(float(l.detach()), 4))
    print("eager losses (every 4th of 40 more steps):", ls)
if len(sys.argv) > 2 and sys.argv[2] == "graph":
    # the whole step (Dropout3d masks from the graph-registered generator) as one hipGraph
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g, stream=side):
            static_loss = step()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    l0 = float(static_loss.detach())
    ls = []
    for i in range(10):
        g.replay()
        ls.append(round(float(static_loss.detach()), 4))
    print("graph losses:", ls)
    t = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"vnet acdc bf16 batch {B}: {dt*1e3:.2f} ms/step hipGraph replay, {B/dt:.1f} volumes/s; loss {l0:.4f} -> {float(static_loss):.4f}")
