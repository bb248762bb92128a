"""-m gpu: the data-parallel step on RCCL.  The GPU box has one device, so the process group has ONE rank — the
collectives still run through RCCL on the GPU (bucketed in-place all-reduce launched from the backward hooks) and
the whole step including them must be capturable in a hipGraph, which is what bench.py does for N > 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ddp1", "1", "--size", "64", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline", "--no-roofline", *extra], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), r.stderr


def test_one_rank_rccl_step_is_graph_capturable_and_matches_eager():
    g, err_g = _bench("--graph", "1")
    e, _ = _bench("--graph", "0")
    assert "RCCL" in g["config"]["workload"]
    assert "hipGraph replay" in g["config"]["workload"], err_g[-2000:]     # the capture did not fall back to eager
    assert "hipGraph replay" not in e["config"]["workload"]
    # same data and seeds; the graph run has taken a few more optimizer steps (side-stream warm-up) when the loss is read
    import math
    lg, le = g["config"]["final_loss"], e["config"]["final_loss"]
    assert math.isfinite(lg) and math.isfinite(le) and lg < le + 0.25, (lg, le)
