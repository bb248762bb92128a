"""-m gpu cases added after round 1's last GPU minute (correct on the host-side executor, first hardware run at the round-end
suite): they live in a file that sorts after every other test file so that `pytest -x` reaches the long-standing parity tests
(and the full-size configuration sweep) first."""
import pytest
import torch

from tests import op_checks as oc

pytestmark = pytest.mark.gpu

F32, BF16 = torch.float32, torch.bfloat16


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_mappool_odd_code_count(dev, dtype):
    oc.check_mappool(dev, dtype, N=1, C=40, M=27, dhw=(4, 5, 6))     # bcv map_size [3,3,3]: element-wise (one-wave) backward


def test_training_utils_surface(dev):
    from tests.optim_checks import check_training_utils_surface
    check_training_utils_surface(dev)


def test_medformer_bcv_structure_fp32_matches_reference_golden(dev):
    from tests.medformer_checks import assert_fp32_parity
    print(assert_fp32_parity("medformer_bcv_tiny", dev))


def test_medformer_bcv_structure_bf16_inside_envelope(dev):
    """0.39 per-tensor gradient-norm error on the executor (a 14-class net with InstanceNorm over 8 voxels at the deepest
    level; one small-norm tensor), logits 0.09 / 0.10."""
    from tests.medformer_checks import run_case
    r, g = run_case("medformer_bcv_tiny", dev, "bf16")
    print(r)
    assert r["logits_err"] < 0.4 and r["aux_err"] < 0.4, r
    assert max(abs(a - b) for a, b in zip(r["ce"] + r["dice"], list(g["ce"]) + list(g["dice"]))) < 0.05, r
    assert r["grad_norm_err"] < 1.0, r


def test_resunet_bottleneck_matches_reference_golden(dev):
    from tests.model_checks import assert_fp32_parity, run_case
    print(assert_fp32_parity("resunet_bottleneck_b16", dev, max_flips=2, g_stem_tol=5e-2, grad_tol=0.15, cos_min=0.999))
    r, g = run_case("resunet_bottleneck_b16", dev, "bf16")
    print(r)
    # bf16 on untrained weights with three convs per block and InstanceNorm over 8 voxels at the deepest level: logits 0.73
    # (max-abs / max-abs) on the executor, yet CE 1.5315 vs 1.5256 and Dice 0.7636 vs 0.7615 — the losses are the criterion
    assert r["logits_err"] < 2.0 and abs(r["ce"] - float(g["ce"])) < 0.05 and abs(r["dice"] - float(g["dice"])) < 0.05, r

