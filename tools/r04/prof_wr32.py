#!/usr/bin/env python
"""Phase cycle profile of k_wgrad_r32 (library built with EXTRA=-DCBIM_WR32_PROF).  python tools/r04/prof_wr32.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import cbim_amd
from cbim_amd import ops

dev, dtype = "cuda", torch.bfloat16
SHAPES = [(32, 32, 128), (96, 64, 128), (64, 64, 64), (192, 128, 64), (128, 128, 32)]
if os.environ.get("CB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["CB_SHAPES"].split(",")]
for cin, cout, s in SHAPES:
    a = torch.relu(torch.randn(1, s, s, s, cin, device=dev) * 1.3 + 0.2).to(dtype)
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    for rep in range(2):
        sys.stderr.write(f"--- wgrad {cin}->{cout} @{s} rep {rep}\n"); sys.stderr.flush()
        ops.conv_wgrad(a, None, dy, geom)
        torch.cuda.synchronize()
