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
    # the graph run takes warmup + 2 (side-stream warm-up) + 1 (first replay) steps before the timed ones: give the
    # eager run the same number of optimizer steps so that the two final losses are THE SAME step of the same training
    e, _ = _bench("--graph", "0", "--warmup", "5")
    assert "RCCL" in g["config"]["workload"]
    assert "hipGraph replay" in g["config"]["workload"], err_g[-2000:]     # the capture did not fall back to eager
    assert "hipGraph replay" not in e["config"]["workload"]
    assert g["n_gpus"] == 1 and g["config"]["rccl_ranks"] == 1
    # all 45 gradients of the ResUNet (conv / stem / head weights, head bias; conv1 | shortcut pairs as one tensor) are written
    # by their kernels straight into the all-reduce buckets: nothing is copied
    assert g["config"]["grad_bucket"] == {"written_in_place": 45, "copied": 0}, g["config"]
    # same data, seeds and step count; every reduction in the engine has a fixed order, so replayed and eager launches
    # of the same kernels give the same loss
    import math
    lg, le = g["config"]["final_loss"], e["config"]["final_loss"]
    assert math.isfinite(lg) and math.isfinite(le) and abs(lg - le) <= 1e-4 * max(1.0, abs(le)), (lg, le)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` self-spawns N ranks (train_ddp.py:413); on a box with fewer GPUs it must fail loudly, never
    print an n_gpus line for fewer devices."""
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout, (r.stdout, r.stderr[-500:])


def _bench2(env_extra, *extra):
    env = dict(os.environ, CBIM_BENCH_SHARE_GPU="1", CBIM_BENCH_BACKEND="gloo", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # ONE line, from rank 0
    return json.loads(lines[0]), r.stderr


def test_two_rank_control_flow_of_bench_on_one_gpu():
    """The N > 1 control flow of bench.py on the 1-GPU box (CBIM_BENCH_SHARE_GPU: both ranks on cuda:0, gloo rendezvous — a test
    vehicle, never a bench line): self-spawn through torch.distributed.run, rank-0 broadcast, bucketed exchange from the backward hooks,
    the eager timing taken first, a graph attempt that gloo refuses on every rank -> the all-ranks agreement falls back to eager
    launches, the roofline's eager steps run on EVERY rank (they hold collectives), one JSON line."""
    d, err = _bench2({})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2
    assert d["config"]["graph"] is False and "eager_ms_per_step" in d["config"], (d["config"], err[-1500:])
    assert d["value"] == pytest.approx(2.0 / (d["ms_per_step"] * 1e-3))
    assert d["config"]["grad_bucket"] == {"written_in_place": 45, "copied": 0}
    assert d["config"]["replica_check"] == {"inputs_differ": True, "weights_identical": True, "aug": False}
    assert d["roofline"]["kernel"].startswith(("k_conv3_rw", "k_wgrad_r32")) and 0 < d["roofline"]["frac"] < 1   # (64^3: the wgrad row leads)
    assert "TEST VEHICLE" in d["config"]["workload"]
    import math
    assert math.isfinite(d["config"]["final_loss"])


def test_two_rank_bench_with_device_side_augmentation():
    """BASELINE configs[3] is "DDP + GPU-side augmentation on": `bench.py --gpus 2 --aug 1` through the same two-rank vehicle.  The
    ranks seed their augmentation draws differently (each trains on its own sample, train_ddp.py:60,330) and the averaged
    gradients keep the replicas' weights identical (train_ddp.py:353) — bench.py checks both on the state the timed steps left and
    reports it in config.replica_check."""
    d, err = _bench2({}, "--aug", "1", "--no-roofline")
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2", d["config"]
    rc = d["config"]["replica_check"]
    assert rc["aug"] is True and rc["inputs_differ"] is True and rc["weights_identical"] is True, (rc, err[-1500:])
    assert "on-device augmentation" in d["config"]["workload"]
    import math
    assert math.isfinite(d["config"]["final_loss"])


def test_graph_attempt_that_does_not_return_reports_the_eager_timing():
    """the watchdog of the N > 1 graph attempt: with a 10 ms limit it fires inside the attempt — rank 0 prints the eager line
    (config.note says so), every rank exits 0"""
    d, err = _bench2({"CBIM_BENCH_GRAPH_TIMEOUT": "0.01"}, "--no-roofline")
    assert d["n_gpus"] == 2 and d["config"]["graph"] is False and "did not return" in d["config"]["note"], d["config"]
    assert "reporting the eager timing" in err


def test_stock_ddp_wrapper_on_one_rank_rccl():
    """train_ddp.py:353: DistributedDataParallel(net, device_ids=[gpu], find_unused_parameters=True) over the engine
    module, here on a 1-rank RCCL group: the reducer's hooks fire from the HIP-kernel backward and leave the same
    gradients as the bare module."""
    import socket
    import torch
    import torch.distributed as dist
    import cbim_amd
    from cbim_amd.model.dim3 import UNet
    from cbim_amd.training.losses import DiceCELoss
    dev = torch.device("cuda", 0)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        cbim_amd.set_compute_dtype("bf16")
        torch.manual_seed(4)
        ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
        net = UNet(1, 8, scale=sc, kernel_size=ks, num_classes=4, block="BasicBlock", norm="in").to(dev)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(1, 1, 32, 32, 32, generator=g).to(dev)
        lab = torch.randint(0, 4, (1, 1, 32, 32, 32), generator=g).to(dev)
        crit = DiceCELoss(torch.tensor([0.5, 1.0, 1.0, 1.0])).to(dev)
        crit(net(x), lab).backward()
        ref = [p.grad.clone() for p in net.parameters()]
        net.zero_grad(set_to_none=True)
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], find_unused_parameters=True)
        ema = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], find_unused_parameters=True)   # :358 wraps the EMA net too
        del ema
        for _ in range(2):
            net.zero_grad(set_to_none=True)
            crit(ddp(x), lab).backward()
        torch.cuda.synchronize()
        for p, r in zip(net.parameters(), ref):
            assert p.grad is not None and torch.equal(p.grad, r)
    finally:
        cbim_amd.set_compute_dtype(None)
        dist.destroy_process_group()


def test_bench_line_is_the_same_over_20_and_200_steps():
    """the driver's 20-step timed region is 0.2 s of a 60 s run.  The per-step time must not depend on
    the length of the timed region: `bench.py --steps 200` against `--steps 20` in the same process conditions, within 2 %."""
    def run(steps):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "5",
                            "--no-cpu-baseline", "--no-roofline", "--secondary", "0"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = run(20), run(200)
    assert a["steps"] == 20 and b["steps"] == 200 and a["config"]["graph"] and b["config"]["graph"]
    rel = abs(a["ms_per_step"] - b["ms_per_step"]) / b["ms_per_step"]
    print(f"ms/step over 20 steps {a['ms_per_step']:.3f}, over 200 steps {b['ms_per_step']:.3f} ({rel * 100:.2f} %)")
    from tests.util import record_parity
    record_parity("bench_20_vs_200_steps", {"ms_20": a["ms_per_step"], "ms_200": b["ms_per_step"], "rel": rel})
    assert rel < 0.02, (a["ms_per_step"], b["ms_per_step"])
