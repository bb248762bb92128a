#!/usr/bin/env python
"""Where the time of k_wgrad_r32 goes on one layer shape (default 32->32 @128^3): CBIM_WR32_DBG ablations
(1 no LDS-DMA after the first tile, 2 no contraction loop, 4 no input-fragment reads, 8 no MFMAs) for both wave layouts.
The ablated kernels are compile-time instantiations: build with `make -C cbim-medical-image-segmentation_amd/csrc
EXTRA=-DCBIM_WR32_ABLATE` first (tools/run_wr32_ablate.sh does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbim_amd
from cbim_amd import ops, _lib
dtype = torch.bfloat16
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cin, cout, s = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32x32x128").split("x"))
x = torch.relu(torch.randn(1, s, s, s, cin, device="cuda")).to(dtype)
dy = torch.randn(1, s, s, s, cout, device="cuda").to(dtype)
geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
L = _lib.lib()
for wv in (8, 4):
    L.cbim_wgrad_r32_waves(wv)
    for dbg in ((0, 0, 0) if os.environ.get("WR_ONLY0") else (0, 0, 1, 2, 3, 4, 5, 8, 9, 12, 13)):
        os.environ["CBIM_WR32_DBG"] = str(dbg)
        t = timeit(lambda: ops.conv_wgrad(x, None, dy, geom))
        print(f"waves={wv} dbg={dbg:2d}: {t:7.1f} us  ({gf / t * 1e3:6.0f} TF/s)", flush=True)
os.environ["CBIM_WR32_DBG"] = "0"
