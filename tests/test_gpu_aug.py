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
    from tests.util import record_parity
    record_parity("augmentation_golden_20x24x28", res)


def test_affine_crop_at_the_benchmarked_shape_matches_oracle(dev):
    from tests import aug_checks
    from tests.util import record_parity
    res = aug_checks.affine_crop_headline(dev)
    print(res)
    record_parity("augmentation_affine_168_to_128", res)


def test_resident_dataset_pipeline_matches_oracle(dev):
    from tests import aug_checks
    aug_checks.resident_pipeline(dev)
