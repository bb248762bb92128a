"""-m gpu: per-kernel parity on a real MI355X through the C ABI (libcbim_hip.so)."""
import pytest
import torch

from tests import op_checks as oc

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16


def test_backend_is_hip(dev):
    from cbim_amd import _lib
    assert dev == "cuda" and _lib.backend() == "hip-gfx950"


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_instnorm(dev, dtype):
    oc.check_instnorm(dev, dtype)
    oc.check_instnorm(dev, dtype, N=1, C=72, dhw=(2, 3, 3))
    oc.check_instnorm(dev, dtype, N=1, C=32, dhw=(40, 33, 31))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_maxpool(dev, dtype):
    oc.check_maxpool(dev, dtype)
    oc.check_maxpool(dev, dtype, dhw=(4, 9, 7), scale=(1, 2, 2))
    oc.check_maxpool(dev, dtype, C=64, dhw=(32, 32, 32))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_upcat(dev, dtype):
    oc.check_upcat(dev, dtype)
    oc.check_upcat(dev, dtype, low=(9, 4, 4), hi=(18, 8, 8), skip_first=False)
    oc.check_upcat(dev, dtype, low=(1, 3, 2), hi=(2, 6, 4))
    oc.check_upcat(dev, dtype, Cl=64, Cs=32, low=(16, 16, 16), hi=(32, 32, 32))


@pytest.mark.parametrize("dtype,N,Cin,Cout,dhw,k", [
    (F32, 1, 8, 8, (4, 8, 8), (3, 3, 3)),
    (F32, 2, 20, 40, (5, 9, 11), (3, 3, 3)),
    (BF16, 2, 40, 72, (6, 9, 10), (3, 3, 3)),
    (F32, 1, 8, 16, (5, 8, 8), (2, 3, 3)),
    (BF16, 1, 16, 8, (3, 12, 8), (1, 3, 3)),
    (BF16, 1, 8, 8, (33, 16, 16), (3, 3, 3)),
    (F32, 1, 4, 4, (2, 2, 2), (3, 3, 3)),
    (BF16, 1, 32, 32, (32, 32, 32), (3, 3, 3)),
    (BF16, 1, 96, 32, (16, 32, 32), (3, 3, 3)),
    (BF16, 1, 64, 128, (16, 16, 16), (3, 3, 3)),
    (BF16, 1, 320, 320, (8, 8, 8), (3, 3, 3)),
    (F32, 1, 32, 64, (16, 16, 16), (3, 3, 3)),
])
def test_conv(dev, dtype, N, Cin, Cout, dhw, k):
    oc.check_conv(dev, dtype, N, Cin, Cout, dhw, k)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_stem_head(dev, dtype):
    oc.check_stem_head(dev, dtype)
    oc.check_stem_head(dev, dtype, Cin=1, base=32, K=16, dhw=(16, 32, 32))


def test_loss(dev):
    oc.check_loss(dev)
    oc.check_loss(dev, N=1, C=16, dhw=(32, 32, 32))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_fused_conv1_shortcut_block(dev, dtype):
    oc.check_fused_block(dev, dtype)
