#!/usr/bin/env python
"""Achieved GB/s of the streaming (non-MFMA-bound) kernels of the ResUNet step at the headline size (1 x 128^3, base 32, bf16):
algorithmic bytes (each tensor read / written once) / HIP-event time.  python tools/stream_bench.py [--size 128] [--json out]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbim_amd
from cbim_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--classes", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--json", default="")
ap.add_argument("--only", default="", help="substring of the rows to time (the others are skipped)")
a = ap.parse_args()
dev = torch.device("cuda:0")
BF = torch.bfloat16
n, C, K = a.size, 32, a.classes
S = n ** 3
MB = 1e6


def timeit(fn, iters=a.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


rows = []


def row(name, nbytes, fn):
    if a.only and a.only not in name:
        return
    us = timeit(fn)
    rows.append({"kernel": name, "MB": round(nbytes / MB, 1), "us": round(us, 1), "GBps": round(nbytes / us / 1e3, 0)})
    print(f"{name:58s} {nbytes / MB:8.1f} MB {us:8.1f} us {nbytes / us / 1e3:7.0f} GB/s  ({nbytes / us / 1e3 / 6300 * 100:4.1f} % of 6.3 TB/s)", flush=True)


act = lambda c=C, s=n: torch.randn(1, s, s, s, c, device=dev).to(BF)
x32, g32 = act(), act()
tb = S * C * 2                                       # one bf16 activation tensor at full resolution
st = ops.instnorm_stats(x32)
row("instnorm_stats (k_partial_sums + finalize)", tb, lambda: ops.instnorm_stats(x32))
row("norm_act_fwd (materialise relu(IN(x)))", 2 * tb, lambda: ops.norm_act_fwd(x32, st, ops.ACT["relu"]))
sums = ops.norm_bwd_sums(g32, x32, st, ops.ACT["relu"], masked=True)
row("norm_bwd_sums (2 reads)", 2 * tb, lambda: ops.norm_bwd_sums(g32, x32, st, ops.ACT["relu"], masked=True))
row("norm_bwd_apply (2 reads, 1 write)", 3 * tb, lambda: ops.norm_bwd_apply(g32, x32, st, sums, ops.ACT["relu"], masked=True))
y, idx = ops.maxpool_fwd(x32, (2, 2, 2))
row("maxpool_fwd 2x2x2", tb + tb // 8 + S * C // 8, lambda: ops.maxpool_fwd(x32, (2, 2, 2)))
gy = torch.randn_like(y)
row("maxpool_bwd 2x2x2", tb + tb // 8 + S * C // 8, lambda: ops.maxpool_bwd(gy, idx, tuple(x32.shape), (2, 2, 2)))
# decoder level 64^3 x 64 -> 128^3: [skip 32 | up 64... the headline ResUNet has low = 64 channels at 64^3, skip = 32 at 128^3
low, skip = act(64, n // 2), act(32, n)
lb, sb, ub = low.numel() * 2, skip.numel() * 2, S * 64 * 2
row("up_stats (tile kernel, reads low)", lb, lambda: ops.up_stats(low, (n, n, n)))
stc = torch.cat([st, ops.up_stats(low, (n, n, n))], 1).contiguous()
row("upcat_act_fwd (reads low + skip, writes 96 ch)", lb + sb + sb + ub, lambda: ops.upcat_act_fwd(low, skip, stc, ops.ACT["relu"], True))
g96 = act(96, n)
cat = ops.upcat_fwd(low, skip, True)
sums96 = ops.norm_bwd_sums(g96, cat, stc, 0, masked=False)
del cat
row("upcat_norm_bwd (reads g, low, skip; writes dskip, dup, dlow)", (sb + ub) + lb + sb + sb + ub + (ub + ub // 2 + ub // 4 + ub // 4 + lb),
    lambda: ops.upcat_norm_bwd(g96, low, skip, stc, sums96, True))
dup = act(64, n)
row("up_adjoint (3 passes: dup -> dlow)", ub + ub // 2 + ub // 2 + ub // 4 + ub // 4 + lb, lambda: ops.up_adjoint(dup, 0, 64, (n // 2,) * 3))
del g96, dup, low, skip
# stem / head
xin = torch.randn(1, 1, n, n, n, device=dev)
w = torch.randn(32, 1, 3, 3, 3, device=dev) * 0.2
for flag in (0, 1):
    _lib.lib().cbim_stem_mfma_enable(flag)
    tag = "matrix cores" if flag else "VALU"
    row(f"stem_fwd (1 -> 32, 3^3) {tag}", S * 4 + tb, lambda: ops.stem_fwd(xin, w, (1, 1, 1), BF))
    row(f"stem_wgrad {tag}", S * 4 + tb, lambda: ops.stem_wgrad(xin, g32, tuple(w.shape), (1, 1, 1)))
wh = torch.randn(K, C, device=dev) * 0.3
bh = torch.randn(K, device=dev)
row(f"head_fwd (32 -> {K}, fp32 planes out)", tb + S * K * 4, lambda: ops.head_fwd(x32, wh, bh))
dz = torch.randn(1, K, n, n, n, device=dev)
for flag in (0, 1):
    _lib.lib().cbim_head_mfma_enable(flag)
    row(f"head_bwd (dx, dw, db) {'k_head_bwd_mfma' if flag else 'k_head_bwd_k (VALU)'}", tb + S * K * 4 + tb, lambda: ops.head_bwd(x32, wh, dz))
lab = torch.randint(0, K, (1, 1, n, n, n), device=dev)
out, coef = ops.dice_ce_fwd(dz, lab, None)
row("dice_ce_fwd", S * K * 4 + S * 8, lambda: ops.dice_ce_fwd(dz, lab, None))
g2 = torch.tensor([1.0, 1.0], device=dev)
row("dice_ce_bwd", 2 * S * K * 4 + S * 8, lambda: ops.dice_ce_bwd(dz, lab, None, coef, g2))
if a.json:
    json.dump(rows, open(a.json, "w"), indent=1)
