"""The C-ABI library loads and exports every symbol include/omni3d_hip.h declares, and the
ctypes signature table matches the header (no compute calls here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT
from omni3d_amd import lib as L

_TYPE_CODE = [("void*", "p"), ("float*", "p"), ("int*", "p"), ("longlong*", "p"), ("unsignedchar*", "p"),
              ("longlong", "l"), ("float", "f"), ("int", "i")]


def _parse_header():
    text = open(os.path.join(ROOT, "include", "omni3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(omni_\w+)\s*\(([^)]*)\)\s*;", text):
        codes = ""
        for arg in m.group(2).split(","):
            a = arg.replace("const", "").strip()
            a = re.sub(r"\s+\w+$", "", a) if not a.endswith("*") else a  # drop the name
            a = a.replace(" ", "")
            for t, c in _TYPE_CODE:
                if a == t:
                    codes += c
                    break
            else:
                raise AssertionError(f"unparsed argument {arg!r} in {m.group(1)}")
        decls[m.group(1)] = codes
    return decls


def test_header_matches_signature_table():
    decls = _parse_header()
    assert decls == L.SIGNATURES


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
