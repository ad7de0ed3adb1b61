"""Host-side mirror of the reference package `cubercnn` (facebookresearch/omni3d) for the MI355X
hot path: same module paths, registry names, class names, config keys and state-dict keys, with
the arithmetic running in the HIP kernels of omni3d_amd/csrc.  `omni3d_amd.install()` exposes it
as `cubercnn` when the reference package is not importable."""
