"""VNet (config/acdc/vnet_3d.yaml: base 16, crop 16x192x192) training-step time, eager, bf16 — and the per-kernel table."""
import sys, time
sys.path.insert(0, ".")
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
print(f"vnet acdc bf16 batch {B}: {dt*1e3:.2f} ms/step, {B/dt:.1f} volumes/s (16x192x192 crops)")
if hasattr(ops, "PROFILE"):
    pass
