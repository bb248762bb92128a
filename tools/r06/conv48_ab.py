#!/usr/bin/env python
"""Round 6 same-box A/B on SwinUNETR's 48-channel layer shapes: k_conv_igemm (round-1 kernel, normalise-on-load or raw) against
k_conv3_rw48 (48 output channels per workgroup, input used as it is).  python tools/r06/conv48_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import cbim_amd
from cbim_amd import _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev, dtype = "cuda", torch.bfloat16
SHAPES = [(48, 48, 128), (96, 48, 128), (8, 48, 128), (48, 48, 64), (96, 48, 64)]
if os.environ.get("CB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["CB_SHAPES"].split(",")]
L = _lib.lib()
LR = ops.ACT["lrelu"]


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


print(f"# reps={reps}; us (TF/s on 2*S*Cin*Cout*27).  T = k_conv_igemm with lrelu(IN(x)) on load; R = k_conv_igemm on the materialised tensor;")
print("# 48 = k_conv3_rw48 on the materialised tensor; pass = k_norm_act_fwd writing it; dgrad = input gradient masked by the activation")
print(f"{'layer':>18s} {'GF':>7s} | {'fwdT':>7s} {'pass':>6s} {'fwdR':>7s} {'fwd48':>7s} {'48+res':>7s} | {'dgT':>7s} {'dg48':>7s} {'dgraw48':>7s} | fwd48 TF/s  dg48 TF/s")
for cin, cout, s in SHAPES:
    x = torch.randn(1, s, s, s, cin, device=dev).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), LR)
    st = ops.instnorm_stats(x)
    a = ops.norm_act_fwd(x, st, LR)
    wp, wd = ops.pack_weights(w, geom, 0), ops.pack_weights(w, geom, 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    res = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
    t_pass = timeit(lambda: ops.norm_act_fwd(x, st, LR))
    L.cbim_conv_rw48_enable(0)
    t_fT = timeit(lambda: ops.conv_fwd(x, wp, geom, in_stats=st, want_stats=True))
    t_fR = timeit(lambda: ops.conv_fwd(a, wp, geom, want_stats=True))
    t_dT = timeit(lambda: ops.conv_dgrad(dy, wd, geom, mask_x=x, mask_stats=st))
    L.cbim_conv_rw48_enable(1)
    t_f48 = timeit(lambda: ops.conv_fwd(a, wp, geom, want_stats=True))
    k1 = L.cbim_conv3d_last_kernel()
    t_f48r = timeit(lambda: ops.conv_fwd(a, wp, geom, res=res, want_stats=True))
    if cin % 48 == 0:
        t_d48 = timeit(lambda: ops.conv_dgrad(dy, wd, geom, mask_x=a, mask_stats=None))
        k2 = L.cbim_conv3d_last_kernel()
        t_dr48 = timeit(lambda: ops.conv_dgrad(dy, wd, geom))
    else:
        t_d48 = t_dr48 = float("nan"); k2 = -1
    L.cbim_wgrad_r32_enable(0)
    t_wT = timeit(lambda: ops.conv_wgrad(x, st, dy, geom))
    t_wR = timeit(lambda: ops.conv_wgrad(a, None, dy, geom))
    L.cbim_wgrad_r32_enable(1)
    t_w32 = timeit(lambda: ops.conv_wgrad(a, None, dy, geom)) if cin % 16 == 0 else float("nan")
    kw = L.cbim_conv3d_wgrad_last_kernel()
    print(f"   wgrad: k_conv_wgrad on load {t_wT:7.1f}  raw {t_wR:7.1f}  k_wgrad_r32 {t_w32:7.1f} us = {gf / t_w32 * 1e3:7.1f} TF/s (kernel {kw})")
    print(f"{cin:4d}->{cout:4d} @{s:3d}^3 {gf:7.1f} | {t_fT:7.1f} {t_pass:6.1f} {t_fR:7.1f} {t_f48:7.1f} {t_f48r:7.1f} | {t_dT:7.1f} {t_d48:7.1f} {t_dr48:7.1f} | "
          f"{gf / t_f48 * 1e3:9.1f} {gf / t_d48 * 1e3:9.1f}   kernels {k1} {k2}", flush=True)
