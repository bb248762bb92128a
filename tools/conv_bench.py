#!/usr/bin/env python
"""Per-layer microbenchmark of the matrix-core conv kernels on the ResUNet-128^3 layer shapes
(SURVEY.md §8a per-layer table).  Usage: python tools/conv_bench.py [bf16|fp32] [reps] [which]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cbim_amd
from cbim_amd import ops

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "fwd,dgrad,wgrad"
dev = "cuda"
SHAPES = [(32, 32, 128), (96, 32, 128), (32, 64, 64), (64, 64, 64), (192, 64, 64), (64, 128, 32), (128, 128, 32),
          (384, 128, 32), (128, 256, 16), (256, 256, 16), (576, 256, 16), (256, 320, 8), (320, 320, 8)]
if os.environ.get("CB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["CB_SHAPES"].split(",")]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if os.environ.get("CB_R32_MINVOX"):
    from cbim_amd import _lib
    _lib.lib().cbim_conv_r32_min_voxels(int(os.environ["CB_R32_MINVOX"]))
print(f"dtype={dtype} reps={reps} dbg={os.environ.get('CBIM_IGEMM_DBG', '0')} r32={os.environ.get('CBIM_CONV_R32', '2')} minvox={os.environ.get('CB_R32_MINVOX', '-')}")
for cin, cout, s in SHAPES:
    x = torch.randn(1, s, s, s, cin, device=dev).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    st = ops.instnorm_stats(x)
    wp, wd = ops.pack_weights(w, geom, 0), ops.pack_weights(w, geom, 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
    line = f"{cin:4d}->{cout:4d} @{s:3d}^3 {gf:7.1f} GF |"
    if "fwd" in which:
        t = timeit(lambda: ops.conv_fwd(x, wp, geom, in_stats=st, want_stats=True))
        line += f" fwd {t*1e3:8.1f} us {gf/t:7.1f} TF/s |"
    if "dgrad" in which:
        t = timeit(lambda: ops.conv_dgrad(dy, wd, geom, mask_x=x, mask_stats=st))
        line += f" dgrad {t*1e3:8.1f} us {gf/t:7.1f} TF/s |"
    if "wgrad" in which:
        t = timeit(lambda: ops.conv_wgrad(x, st, dy, geom))
        line += f" wgrad {t*1e3:8.1f} us {gf/t:7.1f} TF/s |"
    print(line, flush=True)
