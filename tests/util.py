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
