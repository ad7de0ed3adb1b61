"""Small helpers of the benchmark reports: rows of the committed rocprofv3 PMC summaries (profiles/*.csv, written by
tools/pmc_run.py) and a counter of the multiply-adds the MFMA launchers actually execute in one step."""
import csv
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def profile_counters(csv_name, kernel_substr, grid=None):
    """Row of a committed PMC summary for one kernel (+ the file's sha256).  Counters cannot be collected inside a normal bench
    run (the --pmc passes serialise the kernels), so the bench JSON quotes the tracked table it takes them from."""
    path = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(path):
        return None
    digest = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    rows = [r for r in csv.DictReader(open(path)) if kernel_substr in r["kernel"] and (grid is None or int(float(r["grid"])) == grid)]
    if not rows:
        return None
    r = max(rows, key=lambda q: float(q.get("GRBM_GUI_ACTIVE", "0") or 0))

    def f(k):
        try:
            v = float(r[k])
            return v if v == v else None
        except (KeyError, ValueError):
            return None
    out = {"source": f"profiles/{csv_name}", "sha256_16": digest, "kernel": r["kernel"], "grid": int(float(r["grid"]))}
    for k in ("FETCH_SIZE_x2_MB", "WRITE_SIZE_MB", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
              "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "mfma_flop"):
        out[k] = f(k)
    # cycle-based (DVFS-independent) MFMA utilisation: busy SIMD-cycles / (GUI-active cycles per XCD x 1024 SIMDs); GRBM_GUI_ACTIVE is
    # summed over the 8 XCDs on gfx950 (profiles/README.md)
    if out["SQ_VALU_MFMA_BUSY_CYCLES"] and out["GRBM_GUI_ACTIVE"]:
        out["mfma_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if out["SQ_ACTIVE_INST_VALU"] and out["SQ_WAVE_CYCLES"]:
        out["valu_active_frac_of_wave_cycles"] = out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"]
    return out


class ExecutedFlops:
    """`with ExecutedFlops() as c:` counts 2 x multiply-adds of every MFMA launch made through the C ABI (direct convolutions at
    their implicit-GEMM size, Winograd layers at the size of their point GEMMs, FC GEMMs) -- the EXECUTED flops, as opposed to
    the algorithmic (direct-convolution) count SURVEY.md 8(d) prices the step with."""

    def __enter__(self):
        from . import lib
        self.lib = lib.get()
        self.flops = 0.0
        self.by_entry = {}
        self._orig = self.lib.call

        def call(name, *a):
            f = self._flops(name, a)
            if f:
                self.flops += f
                self.by_entry[name] = self.by_entry.get(name, 0.0) + f
            return self._orig(name, *a)
        self.lib.call = call
        return self

    def __exit__(self, *exc):
        self.lib.call = self._orig
        return False

    @staticmethod
    def _flops(name, a):
        def conv(N, H, W, C, K, R, S, stride, pad):
            OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
            return 2.0 * N * OH * OW * K * R * S * C
        if name in ("omni_conv2d_fwd", "omni_conv2d_fwd_algo"):
            return conv(*a[4:13])
        if name == "omni_conv2d_fwd_stats":
            return conv(*a[3:12])
        if name in ("omni_conv2d_dgrad", "omni_conv2d_dgrad_algo", "omni_conv2d_wgrad", "omni_conv2d_wgrad_algo"):
            return conv(*a[3:12])
        if name in ("omni_gemm_batched_fwd", "omni_gemm_batched_fwd_algo", "omni_gemm_batched_wgrad"):
            batch, M, C, K = a[3:7]
            return 2.0 * batch * M * C * K
        if name == "omni_gemm_engine":
            batch, M, N, K = a[5:9]
            return 2.0 * batch * M * N * K
        if name in ("omni_stem_conv_fwd", "omni_stem_conv_fwd_stats", "omni_stem_conv_wgrad"):
            N, H, W, C, K, R = a[3:9]
            return 2.0 * N * H * W * K * R * R * C
        return 0.0


def host_cores():
    """what the box's host really offers (SURVEY.md 8d asks for the physical core count next to a CPU baseline): logical CPUs the
    OS shows, CPUs this process may run on, physical cores (distinct (package, core id) pairs of /proc/cpuinfo) and sockets"""
    import os
    info = {"logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "physical_cores": None, "sockets": None, "model": None}
    try:
        cores, pk = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pk = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cores.add((pk, line.split(":")[1].strip()))
            elif line.startswith("model name") and info["model"] is None:
                info["model"] = line.split(":")[1].strip()
        if cores:
            info["physical_cores"] = len(cores)
            info["sockets"] = len({c[0] for c in cores})
    except OSError:
        pass
    return info
