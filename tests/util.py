"""Shared helpers for the test-suite (test infrastructure only)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    # name: (in_ch, base_ch, classes, scale, kernel_size, block, spatial, batch, seed)
    "resunet_b2_32": (1, 2, 3, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 32), 2, 2023),
    "resunet_b8_32": (1, 8, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "BasicBlock", (32, 32, 32), 1, 2024),
    "resunet_b8_aniso": (2, 8, 5, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]], [[3, 3, 3]] * 5, "BasicBlock",
                         (8, 48, 32), 1, 2025),
    "unet_single_acdc": (1, 8, 4, [[1, 2, 2], [2, 2, 2], [2, 2, 2], [2, 2, 2]],
                         [[1, 3, 3], [2, 3, 3], [3, 3, 3], [3, 3, 3], [3, 3, 3]], "SingleConv",
                         (16, 32, 32), 1, 2026),
    "resunet_bottleneck_b16": (1, 16, 4, [[2, 2, 2]] * 4, [[3, 3, 3]] * 5, "Bottleneck", (32, 32, 32), 1, 2027),
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_state_dict(name):
    """Rebuild the reference-init weights of a golden case from its seed and verify the
    fingerprint recorded when the REAL reference constructor drew them."""
    from oracle.unet_ref import make_unet_state_dict, state_dict_checksum
    in_ch, base, classes, scale, ks, block, shape, batch, seed = CASES[name]
    g = load_golden(name)
    sd = make_unet_state_dict(in_ch, base, classes, ks, block, seed=seed)
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    chk = state_dict_checksum(sd)
    assert abs(chk - float(g["sd_checksum"])) <= 1e-9 * max(1.0, abs(chk)), (chk, float(g["sd_checksum"]))
    return sd


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def grad_compare(named_got, named_ref, bf16=False, verbose=True):
    """Element-wise comparison of every parameter gradient: per tensor the largest |difference| relative
    to the largest |reference| entry, and the cosine of the two flattened tensors.  Returns (worst relative error, lowest
    cosine, number of tensors within 1e-3, number of tensors, name of the worst tensor).  A permuted, transposed or
    sign-flipped interior gradient fails both numbers, which a norm comparison does not see.
    Tensors whose reference gradient is at rounding-noise level (largest entry below 1e-4 of the largest gradient entry of the
    model — e.g. a bias in front of an InstanceNorm, whose gradient is analytically zero: 3e-16 of the scale in float64) are
    compared in absolute terms against that floor and left out of the cosine."""
    scale = max(float(torch.as_tensor(r).double().abs().max()) for r in named_ref.values())
    floor = 1e-4 * scale
    rows = []
    for k, r in named_ref.items():
        g = torch.as_tensor(named_got[k]).detach().double().cpu().flatten()
        r = torch.as_tensor(r).detach().double().cpu().flatten()
        assert g.shape == r.shape, (k, g.shape, r.shape)
        m = float(r.abs().max())
        e = float((g - r).abs().max() / max(m, floor))
        c = float(torch.dot(g, r) / (g.norm() * r.norm()).clamp_min(1e-300)) if m >= floor else 1.0
        rows.append((e, c, m / scale, k))
    worst = max(rows)
    cos_min = min(r[1] for r in rows)
    if verbose:
        for e, c, m, k in sorted(rows, reverse=True)[:5]:
            print(f"  grad {k}: max|d| / max|ref| {e:.2e}, cosine {c:.6f}, max|ref| / model scale {m:.1e}")
        for e, c, m, k in sorted(rows, key=lambda t: t[1])[:3]:
            print(f"  grad (lowest cosine) {k}: cosine {c:.6f}, max|d| / max|ref| {e:.2e}, max|ref| / model scale {m:.1e}")
    return worst[0], cos_min, sum(r[0] <= 1e-3 for r in rows), len(rows), worst[3]


def record_parity(key, values):
    """Append measured parity margins to the round's parity record (magnitudes, not dots).  On the GPU
    box the file lands in gpurun_out/ (merged back by gpurun); the builder copies it to profiles/."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r04_parity.json")
    data = {}
    if os.path.isfile(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data[key] = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in values.items()}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
