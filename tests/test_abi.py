"""The C-ABI library loads and exports every symbol include/omni3d_hip.h declares, and the
ctypes signature table matches the header (no compute calls here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT
from omni3d_amd import lib as L

def _parse_header():
    return L.parse_header()


def test_header_declares_the_abi():
    decls = _parse_header()
    assert len(decls) >= 6 and all(n.startswith("omni_") for n in decls)
    assert decls["omni_iou_box3d"] == "pipippppp"


def test_product_library_exports_every_symbol():
    path = L.LIB_PATH
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "omni3d_amd", "csrc"), "-j8"])
    dll = ctypes.CDLL(path)  # loads without a GPU; nothing is launched
    for name in _parse_header():
        assert hasattr(dll, name), name


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(L.OmniHipError):
        L.HipLibrary(str(tmp_path / "nope.so"))
