"""Which tensor goes non-finite first under hipGraph replay of the VNet step? variants: train|eval, fused|torch optimizer"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cbim_amd
from cbim_amd.model.dim3 import VNet
from cbim_amd.training.losses import DiceCELoss
from cbim_amd.training.optim import FusedAdamW

dev = torch.device("cuda", 0)
mode, optk, size = sys.argv[1], sys.argv[2], int(sys.argv[3])
torch.manual_seed(5)
net = VNet(1, 4, scale=[[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], baseChans=16).to(dev)
net.train() if mode == "train" else net.eval()
crit = DiceCELoss(torch.tensor([0.5, 1, 1, 1.0])).to(dev)
if optk == "fused":
    opt = FusedAdamW(net.parameters(), lr=1e-3, weight_decay=0.05)
else:
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.05, capturable=True)
x = torch.randn(2, 1, 16, size, size, device=dev)
lab = torch.randint(0, 4, (2, 1, 16, size, size), device=dev)
cbim_amd.set_compute_dtype("bf16")


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(net(x), lab)
    loss.backward()
    opt.step()
    return loss


for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
    step()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g, stream=side):
        static_loss = step()
torch.cuda.current_stream().wait_stream(side)
names = {p: n for n, p in net.named_parameters()}
for i in range(16):
    g.replay()
    torch.cuda.synchronize()
    l = float(static_loss.detach())
    badg = [names[p] for p in net.parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    badp = [names[p] for p in net.parameters() if not bool(torch.isfinite(p).all())]
    badb = [n for n, b in net.named_buffers() if not bool(torch.isfinite(b.float()).all())]
    print(mode, optk, size, "replay", i, "loss", round(l, 4), "bad grads", len(badg), badg[:4], "bad params", len(badp), badp[:3], "bad buffers", badb[:3])
    if badg or badp or l != l:
        break
