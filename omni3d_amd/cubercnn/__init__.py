"""Host-side mirror of the reference package `cubercnn` (facebookresearch/omni3d) for the MI355X
hot path: same module paths, registry names, class names, config keys and state-dict keys, with
the arithmetic running in the HIP kernels of omni3d_amd/csrc.  `omni3d_amd.install()` exposes it
as `cubercnn` when the reference package is not importable."""


def __getattr__(name):
    # `from cubercnn import util, vis, data` (tools/train_net.py:51) without importing them at package import time
    if name in ("util", "vis", "data", "evaluation", "solver", "modeling", "config"):
        import importlib
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
