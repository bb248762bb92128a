#!/usr/bin/env python
"""Round-3 A/B on the ResUNet-128^3 layer shapes: "normalise on load" against "materialise act(IN(x)) once", and the
weight-gradient kernels (k_conv_wgrad transformed / raw, k_wgrad_r32 with 8 / 4 waves).  Same box, same process.
    python tools/conv_ab.py [reps]      (CB_SHAPES=CinxCoutxS,... to pick shapes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cbim_amd
from cbim_amd import _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev, dtype = "cuda", torch.bfloat16
# (Cin, Cout, S): the convolutions of the ResUNet as the engine launches them (conv1 | shortcut fused: Cout doubled)
SHAPES = [(32, 32, 128), (96, 64, 128), (32, 128, 64), (64, 64, 64), (192, 128, 64), (64, 256, 32), (128, 128, 32),
          (384, 256, 32), (128, 512, 16), (256, 256, 16), (576, 512, 16), (256, 640, 8), (320, 320, 8)]
if os.environ.get("CB_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["CB_SHAPES"].split(",")]
L = _lib.lib()
if os.environ.get("CB_R32_MINVOX"):
    L.cbim_conv_r32_min_voxels(int(os.environ["CB_R32_MINVOX"]))      # 0: k_conv3_r32 on every eligible shape


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


print(f"# reps={reps}; times in us; TF/s = algorithmic 2*S*Cin*Cout*27 / time")
print("# fwdT = conv(relu(IN(x))) normalised on load (+statistics); pass = k_norm_act_fwd writing a; fwdR = raw conv of a (+statistics)")
print("# dgT / dgA = masked dgrad, mask from (x, stats) / from a with identity stats; wgT / wgR = k_conv_wgrad transformed / raw; w8 / w4 = k_wgrad_r32 8 / 4 waves")
hdr = f"{'layer':>18} {'GF':>7} | {'fwdT':>7} {'pass':>6} {'fwdR':>7} {'R+pass':>7} | {'dgT':>7} {'dgA':>7} | {'wgT':>7} {'wgR':>7} {'w8':>7} {'w4':>7} | {'fwdR TF/s':>9} {'dgA TF/s':>9} {'wbest TF/s':>10}"
print(hdr)
for cin, cout, s in SHAPES:
    x = (torch.randn(1, s, s, s, cin, device=dev) * 1.3 + 0.2).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    st = ops.instnorm_stats(x)
    ident = torch.zeros((1, cin, 2), device=dev)
    ident[..., 1] = 1.0
    wp, wd = ops.pack_weights(w, geom, 0), ops.pack_weights(w, geom, 1)
    dy = torch.randn(1, s, s, s, cout, device=dev).to(dtype)
    a = ops.norm_act_fwd(x, st, 1)
    gf = 2.0 * s ** 3 * cin * cout * 27 / 1e9
    fT = timeit(lambda: ops.conv_fwd(x, wp, geom, in_stats=st, want_stats=True))
    ps = timeit(lambda: ops.norm_act_fwd(x, st, 1))
    fR = timeit(lambda: ops.conv_fwd(a, wp, geom, in_stats=None, want_stats=True))
    dT = timeit(lambda: ops.conv_dgrad(dy, wd, geom, mask_x=x, mask_stats=st))
    dA = timeit(lambda: ops.conv_dgrad(dy, wd, geom, mask_x=a, mask_stats=ident))
    wT = timeit(lambda: ops.conv_wgrad(x, st, dy, geom))
    # raw input on the old kernel (process-wide switch)
    L.cbim_wgrad_r32_enable(0)
    wR = timeit(lambda: ops.conv_wgrad(a, None, dy, geom))
    L.cbim_wgrad_r32_enable(1)
    L.cbim_wgrad_r32_waves(8)
    w8 = timeit(lambda: ops.conv_wgrad(a, None, dy, geom))
    k8 = L.cbim_conv3d_wgrad_last_kernel()
    L.cbim_wgrad_r32_waves(4)
    w4 = timeit(lambda: ops.conv_wgrad(a, None, dy, geom))
    L.cbim_wgrad_r32_waves(8)
    wb = min(w8, w4) if k8 == 1 else wR
    print(f"{cin:4d}->{cout:4d} @{s:3d}^3 {gf:7.1f} | {fT:7.1f} {ps:6.1f} {fR:7.1f} {fR+ps:7.1f} | {dT:7.1f} {dA:7.1f} | {wT:7.1f} {wR:7.1f} "
          f"{w8 if k8 == 1 else float('nan'):7.1f} {w4 if k8 == 1 else float('nan'):7.1f} | {gf/fR*1e3:9.1f} {gf/dA*1e3:9.1f} {gf/wb*1e3:10.1f}", flush=True)
