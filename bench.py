#!/usr/bin/env python
"""bench.py — volumes/s of the model/dim3 training step (forward + CE/Dice loss + backward
[+ gradient all-reduce] + AdamW) on synthetic 1x1x128^3 volumes, BASELINE.json configs[1]:
3D UNet ResBasicBlock (config/amos_ct/resunet_3d.yaml: base 32, 16 classes), bf16, 1 volume per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline     : matrix-core roofline of the dominant kernel (all launches of that conv kernel in a step),
                 algorithmic FLOPs / HIP-event time measured live on the launch stream
  cpu_baseline : the oracle (torch-CPU restatement of the reference, oracle/) timed on this box's
                 host cores on one 1x1x128^3 volume (rank 0, N=1 only): one warm-up pass + two timed passes, median
  secondary    : (default N=1 run only) configs[2] MedFormer, configs[4] SwinUNETR and configs[3] ResUNet + on-device
                 augmentation timed in the same process for 10 replayed steps each (--secondary 0 skips them)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FWD_FLOPS_128 = 2.592e12          # ResUNet-BasicBlock fwd at 1x1x128^3, 16 classes
FWD_FLOPS_128_SWIN = 1.53e12       # SwinUNETR feature 48, 4x128^3 input (conv+linear+attention)
FWD_FLOPS_128_MEDFORMER = 2.34e12  # MedFormer (AMOS yaml) fwd at 1x1x128^3
MEDFORMER_AMOS = dict(base_chan=32, map_size=[4, 4, 4], conv_block="BasicBlock", conv_num=[2, 1, 0, 0, 0, 1, 2, 2],
                      trans_num=[0, 1, 4, 6, 4, 1, 0, 0], chan_num=[64, 128, 256, 320, 256, 128, 64, 32],
                      num_heads=[1, 4, 8, 10, 8, 4, 1, 1], fusion_depth=2, fusion_dim=320, fusion_heads=10,
                      expansion=4, attn_drop=0., proj_drop=0., proj_type="depthwise", norm="in", act="relu",
                      kernel_size=[[3, 3, 3]] * 5, scale=[[2, 2, 2]] * 4, aux_loss=True)  # config/amos_ct/medformer_3d.yaml
PEAK_BF16_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_F32_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
ALG_BYTES_BF16 = 11.6e9            # SURVEY.md §8d compulsory traffic fwd+bwd (bf16)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--base", type=int, default=32)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", type=int, default=1,
                    help="1: replay the whole step (incl. the RCCL gradient all-reduce for N > 1) from one hipGraph, "
                         "captured after warm-up; falls back to eager launches if the capture fails")
    ap.add_argument("--ddp1", type=int, default=0,
                    help="1 (with --gpus 1): run the data-parallel path on a ONE-rank RCCL group — the bucketed gradient "
                         "all-reduce and its hipGraph capture exercised on a single GPU (tests/, not a bench line)")
    ap.add_argument("--cpu-size", type=int, default=128,
                    help="edge of the CPU-baseline sample volume (default: the real 1x1x128^3 volume, no extrapolation)")
    ap.add_argument("--optim", default="fused", choices=["fused", "torch"],
                    help="fused: cbim_amd FusedAdamW (one multi-tensor launch); torch: torch.optim.AdamW(fused=True)")
    ap.add_argument("--model", default="resunet", choices=["resunet", "medformer", "swin_unetr"],
                    help="resunet = BASELINE configs[1] (the headline); medformer = configs[2] (AMOS yaml, aux loss); swin_unetr = configs[4] (4-modality input, feature 48, 4 classes)")
    ap.add_argument("--secondary", type=int, default=1,
                    help="1 (default; N=1 headline run only): after the headline timing, also time MedFormer (configs[2]), SwinUNETR "
                         "(configs[4]) and --aug 1 (configs[3]) for --secondary-steps replayed steps each and report them in `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--aug", type=int, default=0,
                    help="1: draw each step's volume with the on-device augmentation pipeline (configs[3]): affine "
                         "scale/rotate + centre crop from a (size+40)^3 source, then the intensity ops")
    return ap.parse_args()


def synthetic(batch, classes, size, device, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 1, size, size, size, generator=g).clamp_(-7.4, 2.2)   # AMOS-CT intensity range
    coarse = torch.randint(0, classes, (batch, 1, size // 8, size // 8, size // 8), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=(size,) * 3, mode="nearest").long()
    return x.to(device), lab.to(device)


def _timed_reps(one, size, cores, kind, what):
    """one warm-up pass (a 64^3 volume: thread pool, allocator, oneDNN primitive caches) + two timed passes of
    the real sample; the MEDIAN of the timed passes (= their mean for two) is reported, every pass is listed."""
    warm = one(min(64, size))
    reps = [one(size), one(size)]
    dt = sorted(reps)[len(reps) // 2 - 1] * 0.5 + sorted(reps)[len(reps) // 2] * 0.5
    scale = (128.0 / size) ** 3 if size != 128 else 1.0
    return {"value": 1.0 / (dt * scale), "unit": "volumes/s", "cores": cores, "threads": torch.get_num_threads(), "kind": kind,
            "reps_s": [round(t, 2) for t in reps], "warmup_s": round(warm, 2),
            "sample": f"1 volume 1x1x{size}^3 fwd+loss+bwd, {what}, fp32, torch {torch.__version__} CPU, median of 2 timed passes "
                      f"{dt:.1f} s after one 64^3 warm-up pass" + ("" if size == 128 else f" (scaled x{scale:.2f} to 128^3)")}


def cpu_baseline(args):
    """fwd + loss + bwd of ONE 1x1x128^3 volume on the host cores (bounded sample: 2 x ~20-30 s + a 64^3 warm-up).  When the
    reference checkout is mounted (the build container) its own modules are timed (kind "reference"); on the GPU box, where
    /root/reference does not exist, the oracle - the restatement pinned to it by tests/golden - is (kind "port")."""
    from oracle import loss_ref, medformer_ref, unet_ref
    cores = min(os.cpu_count() or 1, 128)
    torch.set_num_threads(cores)
    if args.model == "resunet" and os.path.isdir("/root/reference/model"):
        r = _reference_cpu_baseline(args, cores)
        if r is not None:
            return r
    ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4

    def one(s):
        x, lab = synthetic(1, args.classes, s, "cpu", 2023)
        w = torch.ones(args.classes)
        w[0] = 0.5
        if args.model == "swin_unetr":
            from cbim_amd.model.dim3 import SwinUNETR   # only as the weight initialiser (reference parameter layout)
            from oracle import swin_unetr_ref
            torch.manual_seed(2023)
            sd = {k: v.detach() for k, v in SwinUNETR((s,) * 3, 4, 4, feature_size=48).state_dict().items()}
            for v in sd.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
            x = torch.cat([x] + [synthetic(1, 4, s, "cpu", 3000 + i)[0] for i in range(3)], 1)
            lab = lab.clamp_(max=3)
            w = w[:4]
            t0 = time.perf_counter()
            loss = loss_ref.ce_dice_loss(swin_unetr_ref.swin_unetr_forward(sd, x), lab, w)
        elif args.model == "medformer":
            from cbim_amd.model.dim3 import MedFormer   # only as the weight initialiser (reference parameter layout)
            torch.manual_seed(2023)
            sd = {k: v.detach().requires_grad_(True) for k, v in MedFormer(1, args.classes, **MEDFORMER_AMOS).state_dict().items()}
            m = MEDFORMER_AMOS
            t0 = time.perf_counter()
            outs = medformer_ref.medformer_forward(sd, x, map_size=m["map_size"], num_heads=m["num_heads"],
                                                   fusion_heads=m["fusion_heads"], fusion_depth=m["fusion_depth"],
                                                   kernel_size=m["kernel_size"], scale=m["scale"], act="relu", aux_loss=True)
            loss = sum(0.5 * loss_ref.ce_dice_loss(o, lab, w) for o in outs)
        else:
            sd = unet_ref.make_unet_state_dict(1, args.base, args.classes, ks, "BasicBlock", seed=2023)
            sd = {k: v.requires_grad_(True) for k, v in sd.items()}
            t0 = time.perf_counter()
            logits = unet_ref.unet_forward(sd, x, scale=sc, kernel_size=ks, block="BasicBlock")
            loss = loss_ref.ce_dice_loss(logits, lab, w)
        loss.backward()
        return time.perf_counter() - t0

    return _timed_reps(one, args.cpu_size, cores, "port", "oracle/ (the CPU restatement of the reference)")


def _reference_cpu_baseline(args, cores):
    """The reference's own UNet + CrossEntropyLoss + DiceLoss on the host cores."""
    # the reference's package __init__ files import the whole model zoo (monai, torchvision: not installed): the
    # import shim of SURVEY.md 8c (also used by tests/golden/make_golden.py) registers the two packages as bare
    # namespaces, so only model/dim3/unet.py and its own imports are executed - unmodified reference code
    import importlib
    import types
    ref = "/root/reference"
    saved = {k: sys.modules.get(k) for k in ("model", "model.dim3", "training", "training.losses")}
    try:
        sys.path.insert(0, ref)
        for name, path in (("model", f"{ref}/model"), ("model.dim3", f"{ref}/model/dim3")):
            pkg = types.ModuleType(name)
            pkg.__path__ = [path]
            sys.modules[name] = pkg
        for missing in ("torchvision", "torchvision.transforms"):
            sys.modules.setdefault(missing, types.ModuleType(missing))
        RefUNet = importlib.import_module("model.dim3.unet").UNet
        RefDice = importlib.import_module("training.losses").DiceLoss
    except Exception:
        return None
    finally:
        if ref in sys.path:
            sys.path.remove(ref)
        for k, v in saved.items():           # this repository has packages of the same names: leave no trace
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    def one(s):
        torch.manual_seed(2023)
        net = RefUNet(1, args.base, scale=[[2, 2, 2]] * 4, kernel_size=[[3, 3, 3]] * 5, num_classes=args.classes,
                      block="BasicBlock", norm="in")
        x, lab = synthetic(1, args.classes, s, "cpu", 2023)
        w = torch.ones(args.classes)
        w[0] = 0.5
        ce, dl = torch.nn.CrossEntropyLoss(weight=w), RefDice()
        t0 = time.perf_counter()
        out = net(x)
        loss = ce(out, lab.squeeze(1)) + dl(out, lab)
        loss.backward()
        return time.perf_counter() - t0

    return _timed_reps(one, args.cpu_size, cores, "reference", "/root/reference model.dim3.unet.UNet + CE + DiceLoss")


def _share_gpu():
    """CBIM_BENCH_SHARE_GPU=1 (tests only, never a bench line): ranks may share a device, rendezvous over CBIM_BENCH_BACKEND=gloo —
    the N > 1 control flow of this script (eager timing, graph agreement, fallback, every-rank roofline steps) on a 1-GPU box."""
    return os.environ.get("CBIM_BENCH_SHARE_GPU", "0") == "1"


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node (one process per GPU,
    RCCL rendezvous on 127.0.0.1) the way the reference's trainer spawns its own workers
    (/root/reference/train_ddp.py:413 mp.spawn, :321 init_process_group) — through torch.distributed.run, i.e. the very
    command line the driver uses.  Fails loudly when the node has fewer GPUs than ranks were asked for."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not _share_gpu():
        sys.exit(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); refusing to report a "
                 f"{args.gpus}-GPU number from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    sys.exit(subprocess.call(cmd, env=env))


def _timed(step, steps, world, dev):
    """`steps` steps between barrier + synchronize pairs; the MAX over ranks -> (seconds, last loss)"""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, loss


def time_model(args, dev, rank, world, on_hang=None):
    """Build the model `args.model` names, warm up, capture the step into a hipGraph (args.graph) and time args.steps steps
    between barrier + synchronize pairs.  Returns a dict: ms per step (max over ranks), the step callables, the model facts.
    N > 1: the eager step is timed FIRST; the graph attempt (RCCL collectives captured with the kernels) then runs under a
    watchdog and behind an all-ranks agreement — a rank whose capture failed takes every rank back to eager launches, and a
    capture / replay that does not return within CBIM_BENCH_GRAPH_TIMEOUT seconds (default 240) makes rank 0 report the eager
    timing through `on_hang` and every rank exit, instead of hanging the node."""
    import cbim_amd
    from cbim_amd import ops
    from cbim_amd.model.dim3 import MedFormer, SwinUNETR, UNet
    from cbim_amd.parallel import GradAllReduce
    from cbim_amd.training.losses import DiceCELoss
    torch.manual_seed(2023)
    ks, sc = [[3, 3, 3]] * 5, [[2, 2, 2]] * 4
    in_ch = 1
    if args.model == "medformer":
        net = MedFormer(1, args.classes, **MEDFORMER_AMOS).to(dev)
    elif args.model == "swin_unetr":
        in_ch, args.classes = 4, 4
        net = SwinUNETR((args.size,) * 3, in_ch, args.classes, feature_size=48).to(dev)
    else:
        net = UNet(1, args.base, scale=sc, kernel_size=ks, num_classes=args.classes, block="BasicBlock", norm="in").to(dev)
    net.train()
    w = torch.ones(args.classes)
    w[0] = 0.5
    crit = DiceCELoss(w).to(dev)
    use_graph = bool(args.graph)
    if args.optim == "fused":
        from cbim_amd.training.optim import FusedAdamW
        opt = FusedAdamW(net.parameters(), lr=6e-4, betas=(0.9, 0.999), weight_decay=0.05, eps=1e-5)
    else:
        opt = torch.optim.AdamW(net.parameters(), lr=6e-4, betas=(0.9, 0.999), weight_decay=0.05, eps=1e-5, fused=True,
                                capturable=use_graph)
    ddp = GradAllReduce(net) if dist.is_initialized() else None
    x, lab = synthetic(1, args.classes, args.size, dev, 2023 + rank)
    if in_ch > 1:
        x = torch.cat([x] + [synthetic(1, args.classes, args.size, dev, 3000 + rank + i)[0] for i in range(in_ch - 1)], 1)
    if args.aug:
        # HBM-resident source volumes with the dataset's affine padding (dataset_amos_ct.py:105-165: affine_pad_size 40),
        # samples built by the HIP augmentation kernels on a side stream, one sample ahead of the training step
        import argparse as _ap
        import numpy as np
        from cbim_amd.training.dataset.resident import DevicePrefetcher, ResidentVolumeDataset
        np.random.seed(2023 + rank)
        vols = [synthetic(1, args.classes, args.size + 40, dev, 2023 + 10 * rank + i) for i in range(4)]
        ds = ResidentVolumeDataset([v[0][0] for v in vols], [v[1][0].to(torch.int8) for v in vols],
                                   _ap.Namespace(training_size=[args.size] * 3, affine_pad_size=[40] * 3, scale=[0.3] * 3,
                                                 rotate=[30] * 3, translate=[0] * 3))
        feeder = DevicePrefetcher(ds)
        # the pipeline draws host-side random parameters (and picks its kernels) every step: it stays OUTSIDE the
        # hipGraph — samples are built eagerly on the prefetcher's side stream while the replay runs and handed to the
        # captured step through two static input buffers (one 8 MB + one 16 MB device copy per step)
        x, lab = (t.clone() for t in feeder.next())

    def draw():
        return feeder.next()

    def step():
        opt.zero_grad(set_to_none=True)
        if args.aug:
            xs, ls = draw()
            x.copy_(xs)
            lab.copy_(ls)
        out = net(x)
        ls = lab
        if isinstance(out, (list, tuple)):      # deep supervision, train.py:207-210 (aux_weight [0.5, 0.5])
            loss = sum(0.5 * crit(o, ls) for o in out)
        else:
            loss = crit(out, ls)
        loss.backward()
        if ddp is not None:
            ddp.synchronize()
        opt.step()
        return loss

    eager_step = step
    for _ in range(args.warmup):
        step()
    # gradient tensors per step that their kernel wrote straight into the all-reduce bucket / that were copied into it
    grad_bucket = None
    if ddp is not None and args.warmup > 0:
        grad_bucket = {"written_in_place": ddp.direct_writes // args.warmup, "copied": ddp.copies // args.warmup}
    facts = {"in_ch": in_ch, "classes": args.classes, "grad_bucket": grad_bucket}
    eager_res, watchdog = None, None
    if world > 1 and use_graph:
        dt_e, loss_e = _timed(eager_step, args.steps, world, dev)
        eager_res = dict(facts, ms=dt_e / args.steps * 1e3, dt=dt_e, graph=False, loss=float(loss_e.item()), ddp=ddp,
                         note="hipGraph capture / replay with RCCL did not return: eager launches reported")
        import threading

        def _hung():
            if rank == 0:
                print("bench.py: the hipGraph attempt did not return; reporting the eager timing", file=sys.stderr, flush=True)
                if on_hang is not None:
                    on_hang(eager_res)
            else:
                time.sleep(3.0)
            os._exit(0 if on_hang is not None else 3)
        watchdog = threading.Timer(float(os.environ.get("CBIM_BENCH_GRAPH_TIMEOUT", "240")), _hung)
        watchdog.daemon = True
        watchdog.start()

    def graph_phase(step, use_graph):
        if use_graph:
            # launch-bound inner loop -> one hipGraph: every kernel of fwd+loss+bwd+AdamW is captured once
            # (all launches go to the capturing stream; buffers come from the graph's private pool).
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            aug_on = args.aug
            try:
                if world > 1 and dist.get_backend() != "nccl":
                    # (the shared-GPU test vehicle: a gloo collective inside a capture takes the process down, it does not raise)
                    raise RuntimeError(f"the {dist.get_backend()} backend cannot be captured into a hipGraph")
                with torch.cuda.stream(side):
                    for _ in range(2):
                        step()
                    opt.zero_grad(set_to_none=True)
                    # the RCCL collectives of the gradient exchange are captured with the kernels (N > 1): every rank
                    # captures and replays the same sequence
                    args.aug = 0                        # the captured step reads the static buffers x / lab
                    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local" if ddp is not None else "global"):
                        static_loss = eager_step()
                    args.aug = aug_on
                torch.cuda.current_stream().wait_stream(side)
            except Exception as e:   # capture refused (e.g. an RCCL build without graph support): eager launches
                print(f"bench.py[rank {rank}]: hipGraph capture failed ({type(e).__name__}: {str(e)[:300]}); timing eager launches",
                      file=sys.stderr, flush=True)
                use_graph = False
                args.aug = aug_on
                try:
                    torch.cuda.synchronize()
                except Exception:
                    pass
                if ddp is not None:
                    ddp.reset()
            if world > 1:
                # every rank replays or none does: a replayed graph holds collectives the other ranks must replay too
                flag = torch.tensor([1.0 if use_graph else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                use_graph = bool(flag.item() > 0.5)
            if use_graph:
                def step():
                    if args.aug:
                        xs, ls = draw()
                        x.copy_(xs)
                        lab.copy_(ls)
                    graph.replay()
                    return static_loss
            else:
                opt.zero_grad(set_to_none=True)
                step = eager_step
            step()
        dt, loss = _timed(step, args.steps, world, dev)
        return use_graph, dt, loss

    try:
        use_graph, dt, loss = graph_phase(step, use_graph)
    except BaseException:
        # N > 1: an error here is most likely a peer whose watchdog fired (the group is gone) or a failed replay — this rank's own
        # watchdog reports the eager timing (rank 0) and exits; without a watchdog (N = 1) the error is the result
        if watchdog is not None:
            print(f"bench.py[rank {rank}]: the graph phase raised; waiting for the watchdog", file=sys.stderr, flush=True)
            watchdog.join()
        raise
    if watchdog is not None:
        watchdog.cancel()
    ms = dt / args.steps * 1e3
    loss_val = float(loss.item())
    res = dict(facts, ms=ms, dt=dt, graph=bool(use_graph), loss=loss_val, eager_step=eager_step, ddp=ddp, keep=(net, opt, crit, x, lab))
    if eager_res is not None:
        res["eager_ms"] = eager_res["ms"]
    if dist.is_initialized() and world > 1:
        # what data parallelism promises (train_ddp.py:60,330,353): every rank trained on ITS OWN sample (own sampler shard / own
        # augmentation draws) and the replicas hold THE SAME weights after the averaged-gradient steps — checked on the state the
        # timed steps left behind, one small all-gather every rank takes part in
        tdev = dev if dist.get_backend() == "nccl" else "cpu"
        chk = torch.tensor([float(x.double().sum()), float(sum(p.detach().double().sum() for p in net.parameters())),
                            float(sum(p.detach().double().abs().sum() for p in net.parameters()))], dtype=torch.float64, device=tdev)
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        lst = [t.cpu() for t in lst]
        res["replica_check"] = {
            "inputs_differ": len({round(float(t[0]), 6) for t in lst}) == world,
            "weights_identical": all(abs(float(t[1] - lst[0][1])) <= 1e-9 * float(lst[0][2]) for t in lst),
            "aug": bool(args.aug)}
    return res


def _release(dev):
    """drop the previous model's graph, buffers and packed-weight table before the next model is built"""
    import gc
    from cbim_amd import ops
    gc.collect()
    ops.PACKED.clear()
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()


def kernel_roofline(eager_step, rank, dtype, reps=3):
    """Per-kernel matrix-core roofline of one eager step: every launch of the convolution kernel families timed with HIP events
    on the launch stream (ops.PROFILE), algorithmic FLOPs / time.  Every rank must call it (the eager steps hold the gradient
    collectives); rank 0 gets (table, per) — the other ranks (None, None)."""
    from cbim_amd import ops
    per = {}
    for _ in range(reps):
        ops.PROFILE = [] if rank == 0 else None
        eager_step()
        torch.cuda.synchronize()
        for name, flops, e0, e1, shape, nbytes in (ops.PROFILE or ()):
            d = per.setdefault(name, [0.0, 0.0, 0, 0.0])
            d[0] += flops
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += nbytes
        ops.PROFILE = None
    if rank != 0 or not per:
        return None, None
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    table = {k: {"launches_per_step": v[2] // reps, "avg_launch_ms": v[1] / v[2] * 1e3,
                 "alg_tflop_per_step": v[0] / reps / 1e12, "achieved_tflops": v[0] / v[1] / 1e12,
                 "frac_of_peak": v[0] / v[1] / 1e12 / peak, "alg_bytes_per_launch": v[3] / v[2]} for k, v in per.items()}
    return table, per


DOMINANT_FAMILIES = ("k_conv_igemm", "k_conv3_r32", "k_conv3_rw", "k_wgrad_r32", "k_conv_wgrad")


def _recorded_traffic(model, size):
    """(per-kernel rows, per-step bytes, file name) of the committed rocprofv3 PMC passes for `model` (rocprofv3 cannot run
    inside this process); empty when no recorded pass exists"""
    names = {"resunet": ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_j_traffic.json",
                         "r01_c_traffic.json"),
             "medformer": ("r06_traffic_medformer.json", "r05_traffic_medformer.json", "r04_traffic_medformer.json"),
             "swin_unetr": ("r06_traffic_swin_unetr.json", "r05_traffic_swin_unetr.json", "r04_traffic_swin_unetr.json")}[model]
    tpath = next((q for q in (os.path.join(ROOT, "profiles", t) for t in names) if os.path.isfile(q)), "")
    if size != 128 or not os.path.isfile(tpath):
        return {}, None, ""
    tj = json.load(open(tpath))
    return tj, (tj.get("_step") or {}).get("hbm_bytes_per_step"), os.path.basename(tpath)


# compulsory HBM bytes of one bf16 step (every conv reads its input once and writes its output once, forward +
# dgrad + wgrad; everything else fused).  ResUNet / MedFormer: SURVEY.md 8d; SwinUNETR ("to be derived"): the
# same rule applied to the monai conv blocks and the trunk's Linears / attention cores (tools/r06/swin_bytes.py: sum(in + out) =
# 2 370 Me -> 4.74 GB forward, 14.22 GB per step)
ALG_BYTES_STEP = {"resunet": ALG_BYTES_BF16, "medformer": 15.9e9, "swin_unetr": 14.22e9}


def secondary(args, dev):
    """BASELINE.json configs[2] (MedFormer), configs[4] (SwinUNETR, per-GPU part) and configs[3] (ResUNet with the on-device
    augmentation pipeline, per-GPU part) timed in THIS process after the headline: `--secondary-steps` replayed steps each
    (same step definition: forward + CE/Dice loss + backward + fused AdamW under hipGraph replay).  A failure of one of them is
    reported in its row and never costs the headline line."""
    import argparse as _ap
    rows = {}
    peak = PEAK_BF16_TFLOPS
    for key, model, aug, fwd in (("medformer", "medformer", 0, FWD_FLOPS_128_MEDFORMER),
                                 ("swin_unetr", "swin_unetr", 0, FWD_FLOPS_128_SWIN),
                                 ("resunet_aug", "resunet", 1, FWD_FLOPS_128)):
        a = _ap.Namespace(**vars(args))
        a.model, a.aug, a.steps, a.warmup = model, aug, args.secondary_steps, 3
        _release(dev)
        t0 = time.perf_counter()
        try:
            rr = time_model(a, dev, 0, 1)
            row = {"ms_per_step": rr["ms"], "volumes_per_s": 1e3 / rr["ms"], "steps": a.steps, "warmup": a.warmup,
                   "graph": rr["graph"], "final_loss": rr["loss"], "step_flops": 3.0 * fwd,
                   "step_frac_mfma": 3.0 * fwd / (rr["ms"] * 1e-3) / 1e12 / peak,
                   "step_frac_hbm": ALG_BYTES_STEP[model] / (rr["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "workload": {"medformer": "configs[2]: 3D MedFormer (amos_ct/medformer_3d.yaml, aux loss), 1x1x128^3, 16 classes",
                                "swin_unetr": "configs[4] per GPU: SwinUNETR feature 48, 1x4x128^3, 4 classes (the five MONAI conv "
                                              "blocks of its oracle are parity-unpinned: monai is not installable here)",
                                "resunet_aug": "configs[3] per GPU: ResUNet 1x1x128^3 + HBM-resident volumes, affine / crop / "
                                               "intensity augmentation on the device, prefetched on a side stream"}[key]}
            # the dominant kernel of THIS model on the same clock: three event-timed eager steps, as for the headline
            if not args.no_roofline:
                table, per = kernel_roofline(rr["eager_step"], 0, args.dtype)
                if per:
                    dom = max((k for k in per if k.startswith(DOMINANT_FAMILIES)), key=lambda k: per[k][1])
                    f, tsec, nl, nby = per[dom]
                    tj, step_traffic, tname = _recorded_traffic(model, args.size)
                    traffic = (tj.get(dom) or {}).get("hbm_bytes_per_launch")
                    row["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": f / tsec / 1e12, "peak": peak, "unit": "TFLOP/s",
                                       "frac": f / tsec / 1e12 / peak, "launches_per_step": nl // 3, "avg_launch_ms": tsec / nl * 1e3,
                                       "share_of_conv_time": tsec / sum(v[1] for v in per.values()),
                                       "alg_bytes_per_launch": nby / nl, "traffic": traffic,
                                       "traffic_ratio": (traffic / (nby / nl)) if traffic else None,
                                       "traffic_source": ("profiles/" + tname) if traffic else None,
                                       "step_traffic_bytes": step_traffic,
                                       "step_traffic_ratio": (step_traffic / ALG_BYTES_STEP[model]) if step_traffic else None,
                                       "kernels": table}
            del rr
        except Exception as e:      # noqa: BLE001 - the row says what happened
            row = {"error": f"{type(e).__name__}: {e}"[:300]}
        row["wall_s"] = round(time.perf_counter() - t0, 1)
        rows[key] = row
    _release(dev)
    return rows


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    if _share_gpu():
        local = local % torch.cuda.device_count()
    assert local < torch.cuda.device_count(), f"rank {rank}: LOCAL_RANK {local} but {torch.cuda.device_count()} GPU(s) visible"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("CBIM_BENCH_BACKEND", "nccl") if _share_gpu() else "nccl"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    elif args.ddp1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    import cbim_amd
    from cbim_amd import _lib, ops
    assert _lib.backend() == "hip-gfx950"
    cbim_amd.set_compute_dtype(args.dtype)

    def build_out(r):
        ms, use_graph, loss_val, in_ch, ddp, grad_bucket = r["ms"], r["graph"], r["loss"], r["in_ch"], r["ddp"], r["grad_bucket"]
        value = world * 1.0 / (ms * 1e-3)             # 1 volume per GPU per step (train_ddp.py:330)
        out = _headline(args, world, ms, value, use_graph, loss_val, in_ch, ddp, grad_bucket)
        if r.get("eager_ms") is not None:
            out["config"]["eager_ms_per_step"] = r["eager_ms"]
        if r.get("note"):
            out["config"]["note"] = r["note"]
        if r.get("replica_check"):
            out["config"]["replica_check"] = r["replica_check"]
        return out

    def on_hang(r):
        _flush_c_stdio()
        print(json.dumps(build_out(r)), flush=True)

    box = [time_model(args, dev, rank, world, on_hang=on_hang)]
    _finish(args, build_out(box[0]), box, rank, world, dev)


def _flush_c_stdio():
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _headline(args, world, ms, value, use_graph, loss_val, in_ch, ddp, grad_bucket):
    out = {
        "metric": "3D volumes/sec (fwd+bwd) at 128^3", "value": value, "unit": "volumes/s",
        "n_gpus": dist.get_world_size() if dist.is_initialized() and not args.ddp1 else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": {"medformer": "3D MedFormer (amos_ct/medformer_3d.yaml, aux loss)",
                                "swin_unetr": "SwinUNETR (feature 48, 4-modality BraTS-style input; MONAI conv blocks of its oracle parity-unpinned)",
                                "resunet": "3D UNet ResBasicBlock (amos_ct/resunet_3d.yaml)"}[args.model]
                               + f", 1x{in_ch}x{args.size}^3 per GPU, {args.classes} classes, fwd+CE/Dice loss+bwd+AdamW step"
                               + (", HBM-resident volumes + on-device augmentation (crop/affine/intensity, dataset_amos_ct recipe) prefetched on a side stream" if args.aug else "")
                               + (" (hipGraph replay)" if use_graph else "")
                               + (f", bucketed in-place grad all-reduce ({'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend() + ', TEST VEHICLE'})" if ddp is not None else ""),
                   "global_batch": world, "parallelism": f"dp{world}", "graph": bool(use_graph),
                   "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 0, "final_loss": loss_val},
    }
    if grad_bucket is not None:
        out["config"]["grad_bucket"] = grad_bucket
    return out


def _finish(args, out, box, rank, world, dev):
    """roofline of the dominant kernel, the secondary block, the CPU baseline; rank 0 prints the line.  `box` = [time_model's result]:
    the only reference to the headline model, dropped before the secondary models are built."""
    from cbim_amd import ops
    r = box.pop()
    ms, eager_step = r["ms"], r["eager_step"]
    # ---- roofline of the dominant kernel: every launch of the conv kernels in one step, HIP events on the launch stream
    # (N > 1: EVERY rank runs the three eager steps — they hold the gradient collectives — rank 0 alone records)
    table = per = None
    if not args.no_roofline:
        table, per = kernel_roofline(eager_step, rank, args.dtype)
    if not args.no_roofline and rank == 0:
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        # the dominant kernel = the MFMA conv kernel with the largest share of the step (forward/dgrad kernels; the wgrad
        # entry sums k_conv_wgrad and its reduce)
        dom = max((k for k in per if k.startswith(DOMINANT_FAMILIES[:4])), key=lambda k: per[k][1])
        f, tsec, nl, nby = per[dom]
        fwd128 = {"medformer": FWD_FLOPS_128_MEDFORMER, "swin_unetr": FWD_FLOPS_128_SWIN,
                  "resunet": FWD_FLOPS_128 * (args.base / 32.0) ** 2}[args.model]
        step_flops = 3.0 * fwd128 * (args.size / 128.0) ** 3
        # HBM bytes per launch of the same kernel family from the committed PMC passes (rocprofv3 cannot run
        # inside this process); null when no recorded pass covers this kernel / dtype / model
        tj, step_traffic, tname = _recorded_traffic(args.model, args.size)
        # (rounds 2-4 recorded the k_conv3_rw row under the name of the kernel it grew out of)
        row = tj.get(dom) or tj.get({"k_conv3_rw<bf16>": "k_conv3_r32<bf16>"}.get(dom, dom)) or {}
        traffic = row.get("hbm_bytes_per_launch")
        tpath = tname
        out["roofline"] = {
            "bound": "mfma", "kernel": dom, "achieved": f / tsec / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": f / tsec / 1e12 / peak, "traffic": traffic,
            "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, gfx950-corrected)" % os.path.basename(tpath)
                              if traffic else None,
            "alg_flops_per_launch": f / nl, "avg_launch_ms": tsec / nl * 1e3,
            "alg_bytes_per_launch": nby / nl, "traffic_ratio": (traffic / (nby / nl)) if traffic else None,
            "step_flops": step_flops, "step_achieved": step_flops / (ms * 1e-3) / 1e12,
            "step_frac_mfma": step_flops / (ms * 1e-3) / 1e12 / peak,
            "step_frac_hbm": (ALG_BYTES_STEP[args.model] * (2 if args.dtype == "fp32" else 1) / (ms * 1e-3) / 1e9) / HBM_PEAK_GBS,
            # HBM bytes of ONE step summed over every kernel of the step (the same committed PMC passes, all kernel families)
            # over the model's compulsory bytes (ALG_BYTES_STEP); null until a pass that covers every kernel is recorded
            "step_traffic_bytes": step_traffic if args.dtype == "bf16" else None,
            "step_traffic_ratio": (step_traffic / ALG_BYTES_STEP[args.model]) if (step_traffic and args.dtype == "bf16") else None,
            "kernels": table,
        }
    # ---- configs[2], [4], [3] on the same clock (same process, same box): 10 replayed steps each -----------------------------
    if (args.secondary and rank == 0 and world == 1 and not dist.is_initialized() and args.model == "resunet" and not args.aug
            and args.size == 128 and args.dtype == "bf16"):
        del r, eager_step
        out["secondary"] = secondary(args, dev)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(args)
    # RCCL prints a version banner through C stdio: buffered when stdout is a pipe / file, it would land AFTER the JSON line at
    # process exit.  Every rank flushes its C buffers, the ranks meet, then rank 0 prints — the JSON line is the last line of stdout.
    _flush_c_stdio()
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
        _flush_c_stdio()


if __name__ == "__main__":
    main()
