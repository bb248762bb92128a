#!/usr/bin/env python
"""bf16 envelope of the `norm: bn | ln` MedFormer (TINY widths) on the GPU: engine bf16 vs the oracle under CPU autocast(bf16), both
against the fp32 oracle, at 32^3 (the golden's input) and at 64^3 / other seeds (fresh random inputs) — how much of the ratio is
the single-sample noise of a max-norm statistic on a net whose deepest BatchNorm sees 8 values.  python tools/r06/norm_envelope.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from functools import partial  # noqa: E402

from oracle.medformer_ref import medformer_forward  # noqa: E402
from tests.golden.make_golden import make_labels  # noqa: E402
from tests.medformer_checks import AUX_WEIGHT, MF_CASES, build  # noqa: E402
from tests.util import bf16_envelope_vs_oracle  # noqa: E402

dev = "cuda"
for name in sys.argv[1:] or ["medformer_bn_tiny", "medformer_ln_tiny", "medformer_tiny_32"]:
    m = MF_CASES[name][2]
    fwd = partial(medformer_forward, map_size=m["map_size"], num_heads=m["num_heads"], fusion_heads=m["fusion_heads"],
                  fusion_depth=m["fusion_depth"], kernel_size=m["kernel_size"], scale=m["scale"], act=m["act"], aux_loss=m["aux_loss"])
    dump = {}
    for size, seed in ((32, None), (32, 11), (32, 12), (32, 13), (32, 14), (32, 15), (32, 16), (32, 17)):
        net, g = build(name, dev)
        if seed is None:
            x, lab = torch.from_numpy(g["x"]), torch.from_numpy(g["label"])
        else:
            gen = torch.Generator().manual_seed(seed)
            x = torch.randn((1, 1, size, size, size), generator=gen).clamp_(-7.4, 2.2)
            lab = make_labels(MF_CASES[name][1], (size,) * 3, 1, gen)
        env, bad = bf16_envelope_vs_oracle(dev, net, fwd, x, lab, torch.from_numpy(g["weight"]), tag=None, loss_weights=AUX_WEIGHT, detail=True)
        dump[str(seed)] = {"env": env, "d_eng": env.pop("d_eng", None), "d_ref": env.pop("d_ref", None)}
        print(f"{name} {size}^3 seed {seed}: logits {env['logits_rel_engine']:.4f} / {env['logits_rel_autocast']:.4f} = "
              f"{env['logits_rel_engine'] / env['logits_rel_autocast']:.2f}x  flips {env['argmax_flips_engine']} / {env['argmax_flips_autocast']}  "
              f"cos worst {env['cos_deficit_worst_ratio']:.2f} ({env['cos_deficit_worst_tensor']})  violations {len(bad)}", flush=True)
    import json
    json.dump(dump, open(os.path.join(ROOT, "gpurun_out", f"r06_z_deficits_{name}.json"), "w"))
