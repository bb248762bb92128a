#!/usr/bin/env python
"""Cycle profile of the loop phases of k_conv_igemm / k_conv_wgrad ("what held the conv kernels at 17 %").

Build the profiling variant first (the counters are compiled out of the product):
    touch cbim-medical-image-segmentation_amd/csrc/conv_{igemm,wgrad}.hip
    make -C cbim-medical-image-segmentation_amd/csrc EXTRA=-DCBIM_IGEMM_PROF
then on the GPU:  python tools/conv_phase_profile.py [igemm|wgrad]
Each launch prints one "[igemm prof] ..." / "[wgrad prof] ..." line to stderr: per-wave cycle totals of every phase of
the stage / tile loop (workgroup gridDim.x/2).  Rebuild without EXTRA afterwards (python __graft_entry__.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cbim_amd  # noqa: F401
from cbim_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "igemm"
dtype = torch.bfloat16
SH = [(32, 32, 128), (64, 64, 64), (192, 64, 64)]
if os.environ.get("CB_SHAPES"):
    SH = [tuple(int(v) for v in t.split("x")) for t in os.environ["CB_SHAPES"].split(",")]
for cin, cout, s in SH:
    x = torch.randn(1, s, s, s, cin, device="cuda").to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    st = ops.instnorm_stats(x)
    if which == "igemm":
        wp = ops.pack_weights(w, geom, 0)
        for _ in range(2):
            ops.conv_fwd(x, wp, geom, in_stats=st, want_stats=True)
    else:
        dy = torch.randn(1, s, s, s, cout, device="cuda").to(dtype)
        for _ in range(2):
            ops.conv_wgrad(x, st, dy, geom)
    torch.cuda.synchronize()
