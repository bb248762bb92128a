#!/usr/bin/env python
"""Round-4 A/B on the ResUNet-128^3 layer shapes: k_conv3_r32 against k_conv3_rw (narrow / wide) for the raw forward
(+ residual + statistics) and the activated-mask dgrad (+ sums), as the engine launches them.  Same box, same process.
    python tools/r04/conv_rw_ab.py [reps]      (CB_SHAPES=CinxCoutxS,... to pick shapes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import cbim_amd
from cbim_amd import _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev, dtype = "cuda", torch.bfloat16
SHAPES = [(32, 32, 128), (96, 64, 128), (32, 128, 64), (64, 64, 64), (192, 128, 64), (64, 256, 32), (128, 128, 32),
          (384, 256, 32), (128, 512, 16), (256, 256, 16), (576, 512, 16)]
if os.environ.get("CB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["CB_SHAPES"].split(",")]
L = _lib.lib()


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


print(f"# reps={reps}; us per launch (kernel id: 0 igemm, 1 r32, 2 rw); fwd = raw conv of a + residual + statistics; dg = dgrad masked by a + sums")
print(f"{'layer':>18} {'GF':>7} | {'fwd old':>9} {'fwd rw':>9} {'fwd wide':>9} | {'dg old':>9} {'dg rw':>9} {'dg wide':>9} | best fwd TF/s  best dg TF/s")
for cin, cout, s in SHAPES:
    a = torch.relu(torch.randn(1, s, s, s, cin, device=dev) * 1.3 + 0.2).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    wp, wd = ops.pack_weights(w, geom, 0), ops.pack_weights(w, geom, 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    res = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
    row = []
    for on, wide in ((0, 0), (1, 0), (1, 1)):
        L.cbim_conv_rw_enable(on, wide)
        t = timeit(lambda: ops.conv_fwd(a, wp, geom, res=res, want_stats=True))
        row.append((t, L.cbim_conv3d_last_kernel()))
    for on, wide in ((0, 0), (1, 0), (1, 1)):
        L.cbim_conv_rw_enable(on, wide)
        t = timeit(lambda: ops.conv_dgrad(dy, wd, geom, mask_x=a, mask_stats=None))
        row.append((t, L.cbim_conv3d_last_kernel()))
    L.cbim_conv_rw_enable(1, 1)
    bf, bd = min(r[0] for r in row[:3]), min(r[0] for r in row[3:])
    cells = " ".join(f"{t:7.1f}/{k}" for t, k in row[:3]) + " | " + " ".join(f"{t:7.1f}/{k}" for t, k in row[3:])
    print(f"{cin:4d}->{cout:4d} @{s:3d}^3 {gf:7.1f} | {cells} | {gf / bf * 1e3:9.1f} {gf / bd * 1e3:9.1f}", flush=True)
