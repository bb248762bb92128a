"""The product library (libcbim_hip.so, cross-compiled for gfx950) loads on a GPU-less host and
exports every symbol include/cbim_hip.h declares.  No kernel is launched."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "cbim-medical-image-segmentation_amd", "libcbim_hip.so")


def _declared():
    src = open(os.path.join(ROOT, "include", "cbim_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cbim_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from cbim_amd._lib import EXPORTS
    assert sorted(EXPORTS) == _declared()


def test_hip_library_builds_and_exports_every_symbol():
    # incremental make: a no-op when the in-tree library is up to date
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    h = ctypes.CDLL(SO)
    for name in _declared():
        assert hasattr(h, name), name
    h.cbim_backend.restype = ctypes.c_char_p
    assert h.cbim_backend() == b"hip-gfx950"
    # -fvisibility=hidden: the header's entry points are ALL the library exports (no internal launcher / warm-up symbol)
    r = subprocess.run(["nm", "-D", "--defined-only", SO], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exported = sorted(l.split()[-1] for l in r.stdout.splitlines() if l.split() and not l.split()[-1].startswith("_"))
    assert exported == _declared(), sorted(set(exported) ^ set(_declared()))


def test_missing_library_is_a_hard_error(tmp_path):
    code = ("import os,sys; os.environ['CBIM_HIP_LIBRARY']=%r; sys.path.insert(0,%r); import cbim_amd; "
            "from cbim_amd import _lib\ntry:\n _lib.lib()\nexcept RuntimeError as e:\n print('RAISED', 'no fallback' in str(e))"
            % (str(tmp_path / "nope.so"), ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "RAISED True" in r.stdout, r.stdout + r.stderr
