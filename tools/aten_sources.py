"""Which Python call sites launch the ATen (non-engine) kernels of one training step?  torch.profiler with stacks over one eager step of
bench.py's model; prints, per ATen kernel family, the call sites inside this package with launch counts and device time.
    python tools/aten_sources.py medformer|swin_unetr|resunet"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "medformer"
    sys.argv = [sys.argv[0], "--model", model, "--graph", "0", "--steps", "1", "--warmup", "3", "--no-roofline", "--no-cpu-baseline",
                "--secondary", "0"]
    args = bench.parse()
    import cbim_amd
    cbim_amd.set_compute_dtype(args.dtype)
    dev = torch.device("cuda", 0)
    r = bench.time_model(args, dev, 0, 1)
    step = r["eager_step"]
    step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    by = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
            continue
        site = next((s for s in e.stack if "cbim" in s or "bench.py" in s), e.stack[0] if e.stack else "?")
        k = (e.name, site.replace(ROOT, ""))
        by[k][0] += 1
        by[k][1] += e.device_time_total
    rows = sorted(by.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in by.values())
    print(f"# {model}: top-level aten ops with device time in one eager step: {sum(v[0] for v in by.values())} calls, {tot / 1e3:.3f} ms")
    for (name, site), (n, t) in rows[:70]:
        print(f"{t:9.1f} us {n:5d}x  {name:32s} {site}")


if __name__ == "__main__":
    main()
