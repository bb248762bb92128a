"""Time one training step (forward + CE/Dice + backward + fused AdamW, eager launches) of any SHIPPED 3D configuration
recorded in tests/golden/shipped_configs.json at its own training_size — e.g. the MedFormer configurations that run
on attn_wide.hip:

    python tools/bench_shipped_config.py acdc/medformer_3d.yaml lits/medformer_3d.yaml --steps 5

Prints one JSON line per configuration.  bench.py stays the contract benchmark (BASELINE.json configs); this is the
measurement tool for the other rows of the reference's config directory."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", type=int, default=0, help="1: replay the step from ONE hipGraph captured after warm-up (as bench.py does)")
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs a GPU (no CPU fallback for the product path)"
    dev = torch.device("cuda", 0)
    import cbim_amd
    from cbim_amd.model.utils import get_model
    from cbim_amd.training.losses import DiceCELoss
    from cbim_amd.training.optim import FusedAdamW
    cbim_amd.set_compute_dtype(a.dtype)
    with open(os.path.join(ROOT, "tests", "golden", "shipped_configs.json")) as f:
        shipped = json.load(f)
    for name in a.configs:
        cfg = shipped[name]["args"]
        torch.manual_seed(2023)
        net = get_model(argparse.Namespace(**cfg)).to(dev).train()
        size = cfg["training_size"]
        g = torch.Generator().manual_seed(2023)
        x = torch.randn((1, cfg["in_chan"], *size), generator=g).clamp_(-7.4, 2.2).to(dev)
        lab = torch.randint(0, cfg["classes"], (1, 1, *size), generator=g).to(dev)
        w = torch.tensor(cfg.get("weight", [1.0] * cfg["classes"]), dtype=torch.float32)
        crit = DiceCELoss(w).to(dev)
        opt = FusedAdamW(net.parameters(), lr=6e-4, betas=(0.9, 0.999), weight_decay=0.05, eps=1e-5)
        aux_w = cfg.get("aux_weight", [0.5, 0.5])

        def step():
            opt.zero_grad(set_to_none=True)
            out = net(x)
            if isinstance(out, (list, tuple)):                     # train.py:206-210
                loss = sum(aw * crit(o, lab) for aw, o in zip(aux_w, out))
            else:
                loss = crit(out, lab)
            loss.backward()
            opt.step()
            return loss

        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        run = step
        if a.graph:          # the launch-bound step as one hipGraph (every kernel of fwd + loss + bwd + AdamW captured once)
            graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(graph, stream=side):
                    static_loss = step()
            torch.cuda.current_stream().wait_stream(side)

            def run():
                graph.replay()
                return static_loss
            run()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(json.dumps({"config": name, "model": cfg["model"], "training_size": size, "dtype": a.dtype, "graph": bool(a.graph),
                          "ms_per_step": round(ms, 2), "volumes_per_s": round(1e3 / ms, 2), "steps": a.steps,
                          "final_loss": round(float(loss.detach()), 5), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}),
              flush=True)
        del net, opt, x, lab
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
