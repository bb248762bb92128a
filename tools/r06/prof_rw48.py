#!/usr/bin/env python
"""Phase cycle profile of k_conv3_rw48 / k_conv3_rw (library built with EXTRA=-DCBIM_RW_PROF).  python tools/r06/prof_rw48.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import cbim_amd
from cbim_amd import _lib, ops

dev, dtype = "cuda", torch.bfloat16
SHAPES = [(48, 48, 128), (96, 48, 128), (96, 64, 128), (64, 64, 128)]
L = _lib.lib()
for cin, cout, s in SHAPES:
    a = torch.nn.functional.leaky_relu(torch.randn(1, s, s, s, cin, device=dev) * 1.3 + 0.2, 0.01).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 2 if cout % 48 == 0 else 1)
    wp, wd = ops.pack_weights(w, geom, 0), ops.pack_weights(w, geom, 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    res = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    for rep in range(2):
        sys.stderr.write(f"--- {cin}->{cout} @{s} fwd(stats) rep {rep}\n"); sys.stderr.flush()
        ops.conv_fwd(a, wp, geom, want_stats=True)
        torch.cuda.synchronize()
    sys.stderr.write(f"--- {cin}->{cout} @{s} fwd(res, stats)\n"); sys.stderr.flush()
    ops.conv_fwd(a, wp, geom, res=res, want_stats=True)
    torch.cuda.synchronize()
    if cin % 48 == 0 or cin % 32 == 0:
        sys.stderr.write(f"--- {cin}->{cout} @{s} dgrad(mask a, sums)\n"); sys.stderr.flush()
        ops.conv_dgrad(dy, wd, geom, mask_x=a, mask_stats=None)
        torch.cuda.synchronize()
