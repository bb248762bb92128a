import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cbim_amd
from cbim_amd import ops
dtype = torch.bfloat16
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = 'dbg %s:' % os.environ.get('CBIM_IGEMM_DBG')
for cin, cout, s in [(32, 32, 128), (64, 64, 64), (192, 64, 64)]:
    x = torch.randn(1, s, s, s, cin, device="cuda").to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    geom = ops.ConvGeom(dtype, 1, (s, s, s), cin, cout, (3, 3, 3), (1, 1, 1), 1)
    st = ops.instnorm_stats(x)
    wp = ops.pack_weights(w, geom, 0)
    t = timeit(lambda: ops.conv_fwd(x, wp, geom, in_stats=st, want_stats=True))
    out += f"  {cin}->{cout}@{s}: {t:.0f} us"
print(out, flush=True)
