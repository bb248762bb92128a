#!/usr/bin/env python
"""Phase cycle profile of k_conv3_rw (library built with EXTRA=-DCBIM_RW_PROF): one launch per variant and shape, the
launcher prints the s_memtime sums of workgroup 0 to stderr.  python tools/r04/prof_rw.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import cbim_amd
from cbim_amd import _lib, ops

dev, dtype = "cuda", torch.bfloat16
SHAPES = [(32, 32, 128), (96, 64, 128), (64, 64, 64), (192, 128, 64)]
if os.environ.get("CB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["CB_SHAPES"].split(",")]
L = _lib.lib()
for cin, cout, s in SHAPES:
    a = torch.relu(torch.randn(1, s, s, s, cin, device=dev) * 1.3 + 0.2).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    wp, wd = ops.pack_weights(w, geom, 0), ops.pack_weights(w, geom, 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    res = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    for wide in (0, 1):
        L.cbim_conv_rw_enable(1, wide)
        for rep in range(2):
            sys.stderr.write(f"--- {cin}->{cout} @{s} wide={wide} fwd(res, stats) rep {rep}\n"); sys.stderr.flush()
            ops.conv_fwd(a, wp, geom, res=res, want_stats=True)
            torch.cuda.synchronize()
        sys.stderr.write(f"--- {cin}->{cout} @{s} wide={wide} fwd(plain)\n"); sys.stderr.flush()
        ops.conv_fwd(a, wp, geom)
        torch.cuda.synchronize()
        sys.stderr.write(f"--- {cin}->{cout} @{s} wide={wide} dgrad(mask a, sums)\n"); sys.stderr.flush()
        ops.conv_dgrad(dy, wd, geom, mask_x=a, mask_stats=None)
        torch.cuda.synchronize()
