import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cbim_amd
from cbim_amd import ops
dtype = torch.bfloat16
for cin, cout, s in [(32, 32, 128), (64, 64, 64), (192, 64, 64)]:
    x = torch.randn(1, s, s, s, cin, device="cuda").to(dtype)
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    st = ops.instnorm_stats(x)
    dy = torch.randn(1, s, s, s, cout, device="cuda").to(dtype)
    for _ in range(2): ops.conv_wgrad(x, st, dy, geom)
    torch.cuda.synchronize()
