"""ctypes binding of the C-ABI product library ``libomni3d_hip.so`` (see include/omni3d_hip.h).

The library is built by ``omni3d_amd/csrc/Makefile`` with hipcc for gfx950.  There is NO CPU
fallback: :func:`get` raises if the shared object is missing, and every op wrapper refuses
non-CUDA tensors.  (The CPU test-suite installs a host-emulated build of the *same kernel
sources* through :func:`_install_for_tests`; that seam exists only for tests/.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMNI_LIB_SUFFIX: an A/B build of the same sources next to the default library (csrc/Makefile SUFFIX=...); unset everywhere but in bench A/B runs
LIB_PATH = os.path.join(_HERE, "libomni3d_hip" + os.environ.get("OMNI_LIB_SUFFIX", "") + ".so")

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_double
_CODES = {"p": _P, "i": _I, "l": _L, "f": _F, "d": _D}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "omni3d_hip.h")


def parse_header(path=HEADER_PATH):
    """name -> argument codes ('p' pointer, 'i' int, 'l' long long, 'f' float, 'd' double) for every
    `int omni_*(...)` declared in include/omni3d_hip.h -- the single source of truth of the ABI."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(omni_\w+)\s*\(([^)]*)\)\s*;", text):
        codes = ""
        for arg in m.group(2).split(","):
            a = " ".join(arg.replace("const", " ").split())
            if "*" in a:
                codes += "p"
            elif a.startswith("long long"):
                codes += "l"
            elif a.startswith("float"):
                codes += "f"
            elif a.startswith("double"):
                codes += "d"
            elif a.startswith("int"):
                codes += "i"
            else:
                raise ValueError(f"unparsed argument {arg!r} in {m.group(1)}")
        decls[m.group(1)] = codes
    return decls


SIGNATURES = parse_header()


class OmniHipError(RuntimeError):
    pass


class HipLibrary:
    def __init__(self, path, emulated=False):
        if not os.path.exists(path):
            raise OmniHipError(
                f"{path} not found: build it with `make -C omni3d_amd/csrc` (hipcc, gfx950). "
                "omni3d_amd has no CPU fallback."
            )
        self.path = path
        self.emulated = emulated
        self._dll = ctypes.CDLL(path)
        self._fn = {}
        for name, codes in SIGNATURES.items():
            fn = getattr(self._dll, name)
            fn.argtypes = [_CODES[c] for c in codes]
            fn.restype = ctypes.c_int
            self._fn[name] = fn

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc != 0:
            raise OmniHipError(f"{name} failed with status {rc}")


_lib = None


def get():
    global _lib
    if _lib is None:
        _lib = HipLibrary(LIB_PATH)
    return _lib


def _install_for_tests(lib):
    """Test seam: route calls to a host-emulated build of the kernels (tests/hipemu)."""
    global _lib
    _lib = lib


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_of(t):
    """Raw hipStream_t of torch's current stream on the tensor's device."""
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def check_device(*tensors):
    lib = get()
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda and not lib.emulated:
            raise OmniHipError("omni3d_amd ops run on the GPU only (got a CPU tensor); there is no CPU path")
        if not t.is_contiguous():
            raise OmniHipError("omni3d_amd ops need contiguous tensors")
    return lib
