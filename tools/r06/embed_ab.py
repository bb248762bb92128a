#!/usr/bin/env python
"""Same-box A/B of ops.WGRAD_EMBED_133 (the (1,3,3) weight gradients of the ACDC-structured configs as the centre plane of a 3x3x3
one on k_wgrad_r32) on the shipped ACDC yaml files: one training step replayed from a hipGraph.  python tools/r06/embed_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CFGS = ["acdc/resunet_3d.yaml", "acdc/unet_3d.yaml", "acdc/unet++_3d.yaml", "acdc/attention_unet_3d.yaml", "acdc/medformer_3d.yaml"]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from cbim_amd import ops
    ops.WGRAD_EMBED_133 = sys.argv[2] == "1"
    sys.argv = [sys.argv[0]] + CFGS + ["--graph", "1", "--steps", "10", "--warmup", "3"]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_shipped_config
    bench_shipped_config.main()
else:
    res = {}
    for flag in ("0", "1", "0", "1"):
        out = subprocess.run([sys.executable, __file__, "child", flag], capture_output=True, text=True).stdout
        for l in out.splitlines():
            if l.startswith("{"):
                d = json.loads(l)
                res.setdefault(d["config"], {}).setdefault(flag, []).append(d["ms_per_step"])
    for c, v in res.items():
        print(f"{c:32s} embed off {v.get('0')}  on {v.get('1')}")
