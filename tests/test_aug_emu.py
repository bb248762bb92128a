"""CPU suite: augmentation — oracle vs the reference goldens, and the HIP ops (host-side executor)."""
import pytest
import torch


def test_oracle_augmentation_matches_reference_golden():
    from oracle import augment_ref
    from tests import aug_checks
    res = aug_checks.run("cpu", A=type("O", (), {k: staticmethod(getattr(augment_ref, k)) for k in dir(augment_ref)
                                             if not k.startswith("_")} | {"mirror": staticmethod(lambda t, axis=0: torch.flip(t, dims=[2 + axis]))}))
    for k, v in res.items():
        if k != "affine_lab_tie_voxels":        # (a count of undecided voxels, not an error)
            assert v < 1e-6, (k, v)


def test_hip_augmentation_matches_reference_golden(dev):
    if dev != "cpu":
        pytest.skip("CPU suite")
    from tests import aug_checks
    res = aug_checks.run(dev)
    print(res)
    aug_checks.check(res)
    aug_checks.fused_crop(dev)
    aug_checks.coordinate_crop(dev)


def test_resident_dataset_pipeline_matches_oracle(dev):
    if dev != "cpu":
        pytest.skip("CPU suite")
    from tests import aug_checks
    aug_checks.resident_pipeline(dev)
