import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libcbim_emu.so")


def _want_emu(config) -> bool:
    expr = config.getoption("-m") or ""
    if "not gpu" in expr:
        return True
    if expr.strip() == "gpu":
        return False
    import torch
    return not torch.cuda.is_available()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    if _want_emu(config) and "CBIM_HIP_LIBRARY" not in os.environ:
        # CPU run: execute the very same kernel sources on the host-side executor (tests/emu).
        r = subprocess.run(["make", "-s", "-j8", "-C", EMU_DIR], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building tests/emu failed:\n" + r.stdout + r.stderr)
        os.environ["CBIM_HIP_LIBRARY"] = EMU_LIB


@pytest.fixture(scope="session")
def dev():
    import cbim_amd
    from cbim_amd import _lib
    return "cpu" if _lib.backend() == "emu" else "cuda"
