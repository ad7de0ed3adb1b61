"""Test seam of the GPU-less CI (tests/test_bench_cli.py): a python process started with this directory on PYTHONPATH and
OMNI_EMULATE=1 routes the C-ABI calls of omni3d_amd to tests/hipemu/libomni3d_emu.so -- the unmodified kernel sources
compiled for the host -- the moment omni3d_amd.lib is imported.  It reaches every rank a launcher starts (the environment
is inherited), so `python bench.py --gpus 2` can be exercised end to end over gloo without a GPU.  The product and bench.py
know nothing of it: without this seam omni3d_amd.lib refuses CPU tensors."""
import os
import sys

if os.environ.get("OMNI_EMULATE") == "1":
    import importlib.abc
    import importlib.machinery
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    class _Loader(importlib.abc.Loader):
        def __init__(self, inner):
            self.inner = inner

        def create_module(self, spec):
            return self.inner.create_module(spec)

        def exec_module(self, module):
            self.inner.exec_module(module)
            module._install_for_tests(module.HipLibrary(os.path.join(_root, "tests", "hipemu", "libomni3d_emu.so"), emulated=True))

    class _Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, name, path=None, target=None):
            if name != "omni3d_amd.lib":
                return None
            spec = importlib.machinery.PathFinder.find_spec(name, path)
            if spec is not None:
                spec.loader = _Loader(spec.loader)
            return spec

    sys.meta_path.insert(0, _Finder())
