"""omni3d_amd -- MI355X-native (gfx950) hot path of Cube R-CNN (facebookresearch/omni3d).

Hand-written HIP kernels behind a C ABI (include/omni3d_hip.h, omni3d_amd/csrc) plus the
host-side mirror of the reference's Detectron2 registry surface.  See DESIGN.md.
"""
__version__ = "0.1.0"


def cpu_quota():
    """CPUs this process may really use: the smaller of its affinity mask and the cgroup's CFS quota (cpu.max of cgroup v2,
    cpu.cfs_quota_us / cpu.cfs_period_us of v1); None when nothing limits it below the visible CPU count."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(math.floor(quota))))
    return n if n < (os.cpu_count() or n) else None


def respect_cpu_quota():
    """Round 6.  torch sizes its intra-op thread pool from the CPUs the OS SHOWS (128 threads on the GPU boxes of this project: 2 x 64
    cores), not from what the container may use (cgroup cpu.max = 16 CPUs there).  Every parallel CPU op of the training loop's host
    side -- the zero-fill and slice copies that pad a ragged batch, the packing of the targets -- then wakes 128 threads that spin
    through the CFS quota, and the LAUNCHING thread is throttled with them: measured 75 ms of host stall in two of three eager
    iterations of the multi-scale loop (24 ms -> 95 ms per iteration; profiles/r06_new_shape_*.txt).  Called from build_optimizer
    and bench.py: caps torch's intra-op threads at the quota unless the user chose a count (OMP_NUM_THREADS / MKL_NUM_THREADS set, or
    OMNI_KEEP_THREADS=1).  -> the thread count in effect."""
    import os
    import torch
    if os.environ.get("OMNI_KEEP_THREADS") == "1" or os.environ.get("OMP_NUM_THREADS") or os.environ.get("MKL_NUM_THREADS"):
        return torch.get_num_threads()
    q = cpu_quota()
    if q is not None and torch.get_num_threads() > q:
        torch.set_num_threads(q)
    return torch.get_num_threads()


def install():
    """Makes the reference's import paths resolve to this package: `import cubercnn...` -> `omni3d_amd.cubercnn...` and the
    `detectron2.*` names the hot path uses -> `omni3d_amd.d2.*`, so code written against the reference (the step loop of
    tools/train_net.py:176-313, demo code, tests) runs on the HIP path without edits.  Only the modules this package ships
    resolve (data / evaluation bookkeeping / vis stay out of scope) and nothing is touched when a real
    `detectron2` or `cubercnn` is importable."""
    import importlib
    import importlib.abc
    import importlib.machinery
    import importlib.util
    import sys
    import types

    pkg = __name__
    if any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        return
    for root in ("cubercnn", "detectron2"):
        if root not in sys.modules and importlib.util.find_spec(root) is not None:
            raise RuntimeError(f"a real `{root}` package is importable; omni3d_amd.install() will not shadow it")
    table = {
        "detectron2.config": pkg + ".d2.config", "detectron2.layers": pkg + ".d2.layers",
        "detectron2.structures": pkg + ".d2.structures", "detectron2.solver": pkg + ".d2.solver",
        "detectron2.utils.events": pkg + ".d2.events", "detectron2.utils.comm": pkg + ".d2.comm",
        "detectron2.utils.registry": pkg + ".d2.registry", "detectron2.modeling": pkg + ".cubercnn.modeling.registries",
        # the names tools/train_net.py itself imports (:11-23)
        "detectron2.checkpoint": pkg + ".d2.checkpoint", "detectron2.data": pkg + ".d2.data", "detectron2.engine": pkg + ".d2.engine",
        "detectron2.utils.logger": pkg + ".d2.logger",
    }
    sys.meta_path.insert(0, _AliasFinder(pkg, table))
    for ns in ("detectron2", "detectron2.utils"):
        m = types.ModuleType(ns)
        m.__path__ = []
        sys.modules.setdefault(ns, m)


import importlib.abc as _abc
import importlib.machinery as _machinery


class _AliasLoader(_abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        import importlib
        return importlib.import_module(self.target)      # the real module object, under its real name

    def exec_module(self, module):
        pass


class _AliasFinder(_abc.MetaPathFinder):
    """`cubercnn[.x]` -> `omni3d_amd.cubercnn[.x]`; a fixed table for the `detectron2.*` names."""

    def __init__(self, pkg, table):
        self.pkg, self.table = pkg, table

    def find_spec(self, fullname, path=None, target=None):
        if fullname == "cubercnn" or fullname.startswith("cubercnn."):
            real = self.pkg + "." + fullname
        elif fullname in self.table:
            real = self.table[fullname]
        else:
            return None
        import importlib.util
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return _machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=True)
