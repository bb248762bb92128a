"""Import alias: ``import cbim_amd`` loads the package that lives in the
(non-identifier) directory ``cbim-medical-image-segmentation_amd/``.

The import system returns ``sys.modules['cbim_amd']`` after executing this
file, so replacing that entry with the real package makes ``cbim_amd`` and all
of its submodules (``cbim_amd.ops``, ``cbim_amd.model.utils`` ...) importable.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "cbim-medical-image-segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    "cbim_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cbim_amd"] = _mod
_spec.loader.exec_module(_mod)
