"""Fixed cost of a k_conv_igemm launch: tiny problems, back-to-back launches, HIP events (us per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cbim_amd import ops
from cbim_amd.ops import ConvGeom

dev = torch.device("cuda", 0)
for (N, D, H, W, Ci, Co, k) in [(1, 8, 8, 8, 32, 32, (1, 1, 1)), (1, 8, 8, 8, 32, 32, (3, 3, 3)), (1, 8, 8, 8, 320, 1280, (1, 1, 1)),
                                (1, 16, 16, 16, 256, 1024, (1, 1, 1)), (1, 32, 32, 32, 128, 512, (1, 1, 1)), (1, 64, 64, 64, 64, 256, (1, 1, 1)),
                                (1, 2, 12, 12, 256, 256, (1, 5, 5)), (1, 4, 24, 24, 128, 128, (1, 5, 5)), (1, 16, 96, 96, 32, 32, (1, 5, 5))]:
    x = torch.randn(N, D, H, W, Ci, device=dev).bfloat16()
    w = torch.randn(Co, Ci, *k, device=dev) * 0.05
    g = ConvGeom(x.dtype, N, (D, H, W), Ci, Co, k, tuple(i // 2 for i in k), 0)
    wp = ops.pack_weights(w, g, 0)
    for _ in range(3):
        y = ops.conv_igemm(g.fwd, x, wp, (N, D, H, W, Co))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 50
    e0.record()
    for _ in range(R):
        y = ops.conv_igemm(g.fwd, x, wp, (N, D, H, W, Co))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    gf = 2.0 * N * D * H * W * Ci * Co * k[0] * k[1] * k[2] / 1e9
    yy = y[0] if isinstance(y, tuple) else y
    mb = (x.numel() + yy.numel()) * 2 / 1e6
    print(f"{(N, D, H, W)} {Ci}->{Co} k{k}: {us:7.1f} us/launch (incl. host)  {gf / us * 1e-3 * 1e3:7.1f} TFLOP/s  {mb / us:6.2f} TB/s... GF {gf:.2f} MB {mb:.1f}")
