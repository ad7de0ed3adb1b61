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


def test_conv_forward_refuses_operands_of_2gib_or_more():
    """conv_fwd_kernel addresses x / w through buffer resources with 32-bit byte offsets -- the launcher must return
    OMNI_ERR_ARG for an operand it cannot address instead of wrapping around.  The argument check runs before anything touches the
    device or the pointers, so the product library can be asked on a box without a GPU (no compute call is made)."""
    import pytest
    from omni3d_amd import lib
    L = lib.HipLibrary(lib.LIB_PATH)
    fake = 1 << 20          # never dereferenced: the size check comes first
    # x: 8 x 4096 x 4096 x 32 floats = 16 GiB
    with pytest.raises(lib.OmniHipError, match="status"):
        L.call("omni_conv2d_fwd", fake, fake, None, fake, 8, 4096, 4096, 32, 32, 3, 3, 1, 1, 32, 32, 0, None)
    # w: K x R x S x C = 16384 x 3 x 3 x 4096 floats = 2.25 GiB (the activations are small)
    with pytest.raises(lib.OmniHipError, match="status"):
        L.call("omni_conv2d_fwd", fake, fake, None, fake, 1, 4, 4, 4096, 16384, 3, 3, 1, 1, 4096, 16384, 0, None)
