#!/usr/bin/env python
"""Cost of the fused pieces of the igemm kernel: input transform on/off, epilogue statistics on/off,
residual on/off (tools/conv_bench.py measures the full op)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbim_amd
from cbim_amd import ops
dtype = torch.bfloat16
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, s in [(32, 32, 128), (64, 64, 64), (96, 64, 128), (256, 256, 16)]:
    x = torch.randn(1, s, s, s, cin, device="cuda").to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    st = ops.instnorm_stats(x)
    wp = ops.pack_weights(w, geom, 0)
    res = torch.randn(1, s, s, s, cout, device="cuda").to(dtype)
    out_shape = (1, s, s, s, cout)
    r = {}
    r["plain"] = timeit(lambda: ops.conv_igemm(geom.fwd, x, wp, out_shape))
    r["+xform"] = timeit(lambda: ops.conv_igemm(geom.fwd, x, wp, out_shape, in_stats=st))
    r["+xform+stats"] = timeit(lambda: ops.conv_igemm(geom.fwd, x, wp, out_shape, in_stats=st, want_partials=True))
    r["+xform+stats+res"] = timeit(lambda: ops.conv_igemm(geom.fwd, x, wp, out_shape, in_stats=st, res=res, want_partials=True))
    r["+mask(dgrad-like)"] = timeit(lambda: ops.conv_igemm(geom.fwd, x, wp, out_shape, mask_x=res, mask_stats=ops.instnorm_stats(res) if False else st[:, :cout].contiguous() if cin >= cout else None, want_partials=True)) if cin >= cout else float("nan")
    gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
    print(f"{cin}->{cout}@{s}^3 {gf:.0f} GF: " + "  ".join(f"{k} {v:.0f}us ({gf/v*1e3:.0f} TF/s)" for k, v in r.items()), flush=True)
