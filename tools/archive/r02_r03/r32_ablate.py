#!/usr/bin/env python
"""Where the time of k_conv3_r32 goes: the 32->32 @128^3 layer in its four call forms under the CBIM_R32_DBG
ablations (1 no halo fetch, 2 no MFMA loop, 4 no epilogue, 8 no epilogue loads, 16 no epilogue stores) and the
k_conv_igemm time of the same call (CBIM threshold knob)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbim_amd
from cbim_amd import ops, _lib
dtype = torch.bfloat16
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cin = cout = 32
s = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(1, s, s, s, cin, device="cuda").to(dtype)
w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
st = ops.instnorm_stats(x)
wp = ops.pack_weights(w, geom, 0)
res = torch.randn(1, s, s, s, cout, device="cuda").to(dtype)
osh = (1, s, s, s, cout)
calls = {
    "plain(dma)": lambda: ops.conv_igemm(geom.fwd, x, wp, osh),
    "dma+stats": lambda: ops.conv_igemm(geom.fwd, x, wp, osh, want_partials=True),
    "xform+stats": lambda: ops.conv_igemm(geom.fwd, x, wp, osh, in_stats=st, want_partials=True),
    "xform+stats+res": lambda: ops.conv_igemm(geom.fwd, x, wp, osh, in_stats=st, res=res, want_partials=True),
    "dma+mask+sums": lambda: ops.conv_igemm(geom.fwd, x, wp, osh, mask_x=res, mask_stats=st, want_partials=True),
}
gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
L = _lib.lib()
L.cbim_conv_r32_min_voxels(1 << 40)
print("k_conv_igemm :", "  ".join(f"{k} {timeit(f):.0f}us" for k, f in calls.items()), flush=True)
L.cbim_conv_r32_min_voxels(262144)
for td, dbg in [(4, d) for d in (0, 1, 2, 4, 8, 16, 6)] + [(8, d) for d in (0, 4, 6, 132, 133)]:
    L.cbim_conv_r32_tile_depth(td)
    os.environ["CBIM_R32_DBG"] = str(dbg)
    print(f"r32 td={td} dbg={dbg:2d}  :", "  ".join(f"{k} {timeit(f):.0f}us ({gf/timeit(f)*1e3:.0f}TF)" for k, f in calls.items()), flush=True)
