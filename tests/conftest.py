import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _make(directory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, directory), "-j8"])


@pytest.fixture(scope="session")
def oracle_lib():
    """C oracle (test infrastructure).  Built on demand; prebuilt .so travels to the GPU box."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path) or os.path.exists("/usr/bin/make"):
        try:
            _make("oracle")
        except Exception:
            if not os.path.exists(path):
                raise
    return ctypes.CDLL(path)


@pytest.fixture(scope="session")
def _emu_handle():
    from omni3d_amd import lib as L
    _make("tests/hipemu")
    return L.HipLibrary(os.path.join(ROOT, "tests", "hipemu", "libomni3d_emu.so"), emulated=True)


@pytest.fixture()
def emu_lib(_emu_handle):
    """Routes omni3d_amd calls to the host-emulated build of the kernels for one test."""
    from omni3d_amd import lib as L
    prev = L._lib
    L._install_for_tests(_emu_handle)
    yield _emu_handle
    L._install_for_tests(prev)


@pytest.fixture()
def hip_lib():
    """The real gfx950 library on a real GPU."""
    import torch
    from omni3d_amd import lib as L
    assert torch.cuda.is_available(), "gpu test without a GPU"
    L._install_for_tests(None)
    return L.get()


@pytest.fixture()
def rng():
    return np.random.default_rng(1234)
