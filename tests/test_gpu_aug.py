"""-m gpu: on-device augmentation parity against the reference goldens through the C ABI."""
import pytest

pytestmark = pytest.mark.gpu


def test_hip_augmentation_matches_reference_golden(dev):
    from tests import aug_checks
    res = aug_checks.run(dev)
    print(res)
    aug_checks.check(res)
    aug_checks.fused_crop(dev)
    aug_checks.coordinate_crop(dev)


def test_resident_dataset_pipeline_matches_oracle(dev):
    from tests import aug_checks
    aug_checks.resident_pipeline(dev)
