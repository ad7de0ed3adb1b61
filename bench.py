#!/usr/bin/env python
"""bench.py -- measures the hot path on MI355X.  Contract: see the repo task statement.

  python bench.py --gpus N --steps K --warmup W [--workload all|train|iou3d|infer]

Prints ONE JSON line on rank 0.  BASELINE.json's metric has two halves, and the default workload (`all`) measures both in one
line: the top-level fields are images/sec of the cubercnn_DLA34_FPN training step (batch 4/GPU, synthetic 512x512 Omni3D-shaped
inputs, configs[1] / configs[2]), the `iou3d` object is configs[4] (100k dt x gt box pairs per GPU through box3d_overlap) with
its own roofline and CPU baseline.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself, one process per GPU, the way the
reference's script does (tools/train_net.py:500-510 `launch(main, args.num_gpus, ...)`): bench.py re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`.  Launched under torchrun by somebody
else (WORLD_SIZE set) it is one rank of that job.  Rank 0 reports `n_gpus` = the world size it observed, plus `ranks_observed`
(an all-reduce of ones) and the device every rank ran on.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = 157.3 TFLOP/s dense


# OMNI_BENCH_DEVICE=cpu exists for the GPU-less CI only (tests/test_bench_cli.py): the kernels then have to come from somewhere
# else than libomni3d_hip.so -- the test environment installs the host-compiled build of the same kernel sources -- and the
# line says "device": "cpu".  bench.py itself never installs anything: without that test seam every launcher refuses CPU tensors.
DEVICE = os.environ.get("OMNI_BENCH_DEVICE", "cuda")
ONE_DEVICE = os.environ.get("OMNI_BENCH_ONE_DEVICE") == "1"      # functional check of N > 1 on a 1-GPU box: all ranks on cuda:0


def spawn_ranks(ngpus):
    """--gpus N without a launcher: start N ranks of this script (one process per GPU over RCCL), hand their exit code back.
    Counterpart of the reference's `launch(main, args.num_gpus, num_machines=1, dist_url=...)`, tools/train_net.py:500-510."""
    import socket
    import subprocess
    if DEVICE == "cuda" and not ONE_DEVICE and torch.cuda.device_count() < ngpus:
        sys.exit(f"bench.py --gpus {ngpus}: this node shows {torch.cuda.device_count()} GPU(s); one process per GPU is the measured "
                 "configuration (OMNI_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0 over gloo as a functional check)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    from omni3d_amd import cpu_quota
    env.setdefault("OMP_NUM_THREADS", str(max(1, (cpu_quota() or os.cpu_count() or 1) // ngpus)))      # (what the cgroup grants, shared by the ranks)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def setup_dist(ngpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    info = {"backend": None, "device": DEVICE}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the measured configuration is RCCL ("nccl"), one GPU per rank; ranks that share a device (OMNI_BENCH_ONE_DEVICE=1) or
        # run on the host (OMNI_BENCH_DEVICE=cpu) exchange through gloo
        backend = os.environ.get("OMNI_BENCH_BACKEND", "gloo" if (ONE_DEVICE or DEVICE != "cuda") else "nccl")
        if ONE_DEVICE:
            local = 0
        if DEVICE == "cuda":
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        info["backend"] = "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend
    elif DEVICE == "cuda":
        torch.cuda.set_device(0)
    if ngpus != world:
        raise SystemExit(f"bench.py --gpus {ngpus} but the launcher started {world} rank(s)")
    return world, rank, local, info


def observe_ranks(world, local, info):
    """what the job really was: number of ranks that answer an all-reduce of ones, and the device of each"""
    if world == 1:
        info.update(ranks_observed=1, devices=[f"{DEVICE}:0" if DEVICE == "cuda" else DEVICE])
        return info
    ones = torch.ones(1, device=DEVICE)
    dist.all_reduce(ones)
    devs = [None] * world
    dist.all_gather_object(devs, f"cuda:{local}" if DEVICE == "cuda" else "cpu")
    info.update(ranks_observed=int(ones.item()), devices=devs, one_gpu_per_rank=DEVICE == "cuda" and len(set(devs)) == world)
    return info


def dev_sync():
    if DEVICE == "cuda":
        torch.cuda.synchronize()


def barrier_sync(world):
    dev_sync()
    if world > 1:
        dist.barrier()
    dev_sync()


def max_over_ranks(x, world):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=DEVICE)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


from omni3d_amd.profile_io import latest_profile, profile_counters  # noqa: E402


class _Timer:
    """HIP events on torch's current stream (the stream the launchers use); host clock when the kernels run emulated on the CPU"""
    def __init__(self):
        self.cuda = DEVICE == "cuda"
        self.a, self.b = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.cuda else (None, None)

    def start(self):
        if self.cuda:
            self.a.record()
        else:
            self.t0 = time.perf_counter()

    def stop(self):
        if self.cuda:
            self.b.record()
        else:
            self.t1 = time.perf_counter()

    def ms(self):
        return self.a.elapsed_time(self.b) if self.cuda else 1e3 * (self.t1 - self.t0)


IOU_PAIRS = int(os.environ.get("OMNI_BENCH_IOU_PAIRS", "100000"))      # BASELINE configs[4]: 100k; smaller only for the emulated CI run


def run_iou3d(args, world, rank):
    """One step = one pass of box3d_overlap's work over 100k (dt, gt) pairs resident in HBM.
    Pairs are independent, so ranks shard them with no collective (weak scaling)."""
    from omni3d_amd import boxgen
    from omni3d_amd.kernels import iou3d
    P = IOU_PAIRS
    rng = np.random.default_rng(1000 + rank)
    dt, gt, _ = boxgen.omni3d_like_pairs(rng, P)
    d, g = torch.from_numpy(dt).to(DEVICE), torch.from_numpy(gt).to(DEVICE)
    ar = torch.arange(P, dtype=torch.int32, device=DEVICE)

    def step():
        valid, _ = iou3d.box3d_validity(d)
        return iou3d.iou_box3d_pairs(d, g, ar, ar, valid1=valid)[1]

    for _ in range(args.warmup):
        step()
    ev = [_Timer() for _ in range(args.steps)]
    barrier_sync(world)
    t0 = time.perf_counter()
    for e in ev:
        valid, _ = iou3d.box3d_validity(d)
        e.start()
        out = iou3d.iou_box3d_pairs(d, g, ar, ar, valid1=valid)[1]
        e.stop()
    barrier_sync(world)
    dt_s = max_over_ranks(time.perf_counter() - t0, world)
    kern_ms = float(np.mean([e.ms() for e in ev]))
    alg_bytes = P * (192 + 4)
    pmc = profile_counters(IOU_PMC, "iou_box3d_kernel")
    res = {
        "metric": "IoU3D box pairs/sec (box3d_overlap, 100k dt x gt pairs)", "value": P * world * args.steps / dt_s,
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt_s / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"iou3d: {P} Omni3D-shaped oriented box pairs per GPU (50% overlapping, 1% degenerate)",
                   "pairs_per_gpu": P, "parallelism": f"pairs sharded x{world}, no collective"},
        # IoU3D is neither HBM- nor MFMA-bound (196 B of I/O against ~1e4 branchy scalar flops per pair): the
        # required `roofline` object carries the HBM sanity bound, `valu` the issue-side utilisation from the committed PMC pass
        "roofline": {"bound": "hbm", "kernel": "iou_box3d_kernel", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": (pmc["FETCH_SIZE_x2_MB"] + pmc["WRITE_SIZE_MB"]) * 1e6 if pmc and pmc.get("FETCH_SIZE_x2_MB") and pmc.get("WRITE_SIZE_MB") else None,
                     "kernel_ms": kern_ms, "pairs_per_s_kernel": P / (kern_ms * 1e-3),
                     # the kernel's own figure: it is VALU-issue bound, so the number to push is the share of SIMD-cycles that have a VALU
                     # instruction in flight (committed PMC pass; SQ_* in quad-cycles) -- `frac` above is only the schema's HBM sanity bound
                     "kernel_bound": "valu-issue",
                     "valu_busy_frac_of_simd_cycles": pmc.get("valu_busy_frac_of_simd_cycles") if pmc else None,
                     "valu_lane_utilisation": pmc.get("valu_lane_utilisation") if pmc else None,
                     "note": "196 B/pair algorithmic I/O; the kernel is VALU / branch bound, not HBM bound: `valu_busy_frac_of_simd_cycles` = "
                             "SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) from the committed PMC pass in `valu`",
                     "valu": pmc},
    }
    if rank == 0:       # the CPU leg is reported at N = 1 only (the other ranks would wait on it)
        try:
            res["cpu_baseline"] = cpu_baseline_iou3d(dt, gt) if world == 1 else {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port",
                                                                                   "sample": "reported at N=1 only"}
        except Exception as e:  # noqa: BLE001 -- a failing CPU leg must never take the measured line down with it
            res["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {str(e)[:200]}"}
    return res


IOU_PMC = latest_profile("pmc_iou3d.csv")


def cpu_baseline_iou3d(dt, gt, nsample=20000):
    """the restated pytorch3d algorithm (oracle/iou_box3d_oracle.c) single-thread, OpenMP on all host cores, and
    the PYTHON-LOOP form the reference really runs (omni3d_evaluation.py:1339-1343: one box3d_overlap call per (image, category)
    group from a dict comprehension) -- each on a bounded sample of the same pairs."""
    if os.environ.get("OMNI_BENCH_SKIP_CPU") == "1":       # profiling runs: do not spend GPU-box minutes on the CPU leg
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "skipped (OMNI_BENCH_SKIP_CPU=1)"}
    nsample = min(nsample, len(dt))
    from omni3d_amd import cpu_quota
    cores = min(cpu_quota() or len(os.sched_getaffinity(0)), 64)          # OpenMP team of the all-cores leg (set before libgomp starts): what the cgroup grants
    from omni3d_amd.profile_io import host_cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    Pt = ctypes.c_void_p
    out = np.zeros(nsample, np.float32)
    a, b = np.ascontiguousarray(dt[:nsample]), np.ascontiguousarray(gt[:nsample])

    def timed(fn, budget):
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget:
            fn()
            reps += 1
        return reps, time.perf_counter() - t0
    reps, el = timed(lambda: orc.iou_box3d_pairs_oracle(a.ctypes.data_as(Pt), b.ctypes.data_as(Pt), nsample, out.ctypes.data_as(Pt)), 8.0)
    single = nsample * reps / el
    reps, el = timed(lambda: orc.iou_box3d_pairs_oracle_omp(a.ctypes.data_as(Pt), b.ctypes.data_as(Pt), nsample, out.ctypes.data_as(Pt)), 5.0)
    omp = nsample * reps / el
    # python-loop form: groups of <= 100 dt x U{1..8} gt like an evaluation, one call + tensor construction per group
    rs = np.random.RandomState(0)
    groups, pos = [], 0
    while pos < nsample:
        nd, ng = int(rs.randint(10, 101)), int(rs.randint(1, 9))
        groups.append((a[pos:pos + nd].tolist(), b[pos:pos + ng].tolist()))
        pos += nd
    npairs = sum(len(x) * len(y) for x, y in groups)

    def loop():
        for dl, gl in groups:
            dd, gg = torch.tensor(dl, dtype=torch.float32), torch.tensor(gl, dtype=torch.float32)      # :1412-1413
            n, m = dd.shape[0], gg.shape[0]
            o = np.empty((n, m), np.float32)
            orc.box3d_overlap_oracle(dd.numpy().ctypes.data_as(Pt), n, gg.numpy().ctypes.data_as(Pt), m, ctypes.c_float(1e-4),
                                     ctypes.c_float(1e-8), o.ctypes.data_as(Pt))
    reps, el = timed(loop, 8.0)
    pyloop = npairs * reps / el
    return {"value": single, "unit": "pairs/s", "cores": 1, "kind": "port", "host": host_cores(),
            "sample": f"first {nsample} pairs of the same workload, oracle/iou_box3d_oracle.c, 1 thread, ~8 s",
            "openmp": {"value": omp, "cores": cores, "sample": f"same {nsample} pairs, #pragma omp parallel for, {cores} threads (the box's cgroup may grant fewer "
                                                                  "CPUs than it shows: the measured speed-up over 1 thread is what it is), ~5 s"},
            "python_loop": {"value": pyloop, "cores": 1,
                            "sample": f"{len(groups)} evaluator-shaped groups (<=100 dt x 1-8 gt, {npairs} pairs): torch.tensor + one "
                                      "box3d_overlap call per group like omni3d_evaluation.py:1339-1343, ~8 s"}}


def extra_leg(argv, env):
    """one more workload of this script in its own process (another model configuration is a module-level constant of bench_train);
    -> the fields of its JSON line a reader needs, or the error"""
    import subprocess
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "config")
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1"] + argv, env=dict(os.environ, **env), capture_output=True,
                             text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        return {k: r[k] for k in keep if k in r}
    except Exception as e:  # noqa: BLE001 -- an extra leg must never take the measured line down with it
        return {"value": None, "error": f"{type(e).__name__}: {str(e)[:200]}"}


LINE_CAP = 6000          # bytes of the ONE stdout line (the 21 KB line of round 5 did not survive the driver's parser)
DETAIL_NAME = "bench_detail.json"


def write_detail(res):
    """Everything the line used to carry (17 kernel families, HBM-bound kernels, the drop-in loop legs, PMC rows, the N > 1 `exchange`
    object) goes to bench_detail.json beside this script; the line quotes its sha256.  -> {"file", "sha256_16", "bytes"} or the error"""
    import hashlib
    text = json.dumps(res, indent=1, default=str)
    for d in (os.environ.get("OMNI_BENCH_DETAIL_DIR") or ROOT, "/tmp"):
        try:
            path = os.path.join(d, DETAIL_NAME)
            with open(path, "w") as f:
                f.write(text)
            return {"file": path if d != ROOT else DETAIL_NAME, "sha256_16": hashlib.sha256(text.encode()).hexdigest()[:16], "bytes": len(text)}
        except OSError as e:
            err = f"{type(e).__name__}: {e}"
    return {"file": None, "error": err[:120]}


def _r(x, nd=4):
    """numbers of the line with the digits a reader uses"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}") if abs(x) < 1 else round(x, nd)
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(res, detail):
    """The stdout line: the contract's top-level fields, `roofline` and `cpu_baseline` of the dominant kernel / the CPU port, the
    self-diagnosis of the step time (five windows, clocks / power / temperature under load, per-stage ends), and ONE number for each
    side leg.  Hard cap LINE_CAP bytes -- optional parts are dropped (and named in `dropped`) before the cap is ever exceeded;
    tests/test_bench_cli.py asserts it."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: res.get(k) for k in top}
    cfg = dict(res.get("config") or {})
    line["config"] = cfg
    if "windows" in res:
        line["windows"] = _pick(res["windows"], ("ms_per_step", "min", "median", "max", "value_is", "conditioning_steps_before_warmup"))
    gs = res.get("gpu_state")
    if gs:
        line["gpu_state"] = {"idle": gs.get("idle"), "under_load": gs.get("under_load"), "neighbours": gs.get("neighbours")}
    if res.get("stage_ends"):
        line["stage_ends"] = res["stage_ends"]
    line.update(_pick(res, ("host_enqueue_ms_per_step", "step_mfma_frac", "step_executed_mfma_frac", "step_executed_gflop", "loss_first_last",
                            "skipped_steps", "functional_check_only")))
    if "launch_mode" in res:
        line["launch_mode"] = str(res["launch_mode"])[:100]
    rf = res.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_symbol_weighted", "traffic",
                                      "algorithmic_bytes_per_launch", "kernel_ms", "pmc_mfma_busy_frac", "operands", "table"))
        if "kernel" in line["roofline"]:
            line["roofline"]["kernel"] = line["roofline"]["kernel"][:160]
        worst = [f for f in rf.get("families", []) if f.get("in_step")]
        if worst:       # the family furthest below the roofline INSIDE the step (the full table is in the detail file)
            w = min(worst, key=lambda f: f["in_step"]["frac"])
            line["roofline"]["furthest_family_in_step"] = {"kernel": w["kernel"][:60], "frac_in_step": w["in_step"]["frac"], "frac_alone": w["frac"]}
        line["roofline"]["families_in_detail"] = len(rf.get("families", []))
    else:
        line["roofline"] = rf
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:200]
        host = cb.get("host") or {}
        if host:
            line["cpu_baseline"]["host"] = "%s, %s physical cores" % (host.get("model"), host.get("physical_cores"))
    else:
        line["cpu_baseline"] = cb
    io = res.get("iou3d")
    if isinstance(io, dict):
        r3 = io.get("roofline") or {}
        c3 = io.get("cpu_baseline") or {}
        line["iou3d"] = {"metric": io.get("metric"), "value": io.get("value"), "unit": io.get("unit"), "ms_per_step": io.get("ms_per_step"),
                         "workload": (io.get("config") or {}).get("workload"),
                         "roofline": _pick(r3, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "kernel_bound",
                                                "valu_busy_frac_of_simd_cycles", "valu_lane_utilisation")),
                         "cpu_baseline": dict(_pick(c3, ("value", "unit", "cores", "kind")), sample=str(c3.get("sample", ""))[:120],
                                              openmp_value=(c3.get("openmp") or {}).get("value"), openmp_cores=(c3.get("openmp") or {}).get("cores"))}
    for leg in ("infer", "resnet34", "train_split6"):
        if isinstance(res.get(leg), dict):
            line[leg] = _pick(res[leg], ("value", "unit", "ms_per_step", "error"))
    if res.get("dropin_loop_ms_per_step") is not None:
        line["dropin_loop"] = {"ms_per_step": res["dropin_loop_ms_per_step"]}
        ms = res.get("dropin_loop_multiscale_stream") or {}
        if ms.get("ms_per_iteration_whole_region") is not None:
            line["dropin_loop"]["multiscale_stream"] = _pick(ms, ("iterations", "ms_per_iteration_whole_region", "ms_per_iteration_last_quarter",
                                                                   "fixed_shape_step_scaled_by_pixels_ms", "whole_region_vs_pixel_scaled_fixed_shape", "hit_rate", "eager_new_shape_ms",
                                                                   "capture_ms", "replay_ms"))
    sp = res.get("bf16_split")
    if isinstance(sp, dict) and sp.get("rows"):        # the opt-in experiment: one row, four numbers (the rest in the detail file)
        r0 = sp["rows"][0]
        line["bf16_split_experiment"] = {"shape": r0["shape"], "fp32_mfma_ms": r0["fp32_mfma"]["kernel_ms"], "split6_ms": r0["split6"]["kernel_ms"],
                                         "split3_ms": r0["split3"]["kernel_ms"], "split6_err_over_fp32_err": r0["split6_err_over_fp32_err"],
                                         "split3_err_over_fp32_err": r0["split3_err_over_fp32_err"], "on_measured_path": False}
    if "nonstandard" in res:
        line["nonstandard"] = _pick(res["nonstandard"], ("ims_per_gpu", "image_size", "device"))
    ex = res.get("exchange")
    if isinstance(ex, dict):       # N > 1: two numbers, the rest (per-call table, stage timeline, env) in the detail file
        line["exchange"] = _pick(ex, ("exposed_ms", "all_reduce_calls_per_step", "bytes_per_step", "chunk_mb", "merge_from_stage"))
    la = res.get("launch") or {}
    line["launch"] = _pick(la, ("backend", "device", "ranks_observed", "one_gpu_per_rank", "host_threads"))
    line["detail"] = detail
    line = _r(line)
    dropped = []
    for k in ("launch_mode", "bf16_split_experiment", "nonstandard", "train_split6", "stage_ends", "exchange", "dropin_loop", "resnet34", "infer", "gpu_state", "windows"):
        if len(json.dumps(line)) <= LINE_CAP:
            break
        if k in line:
            dropped.append(k)
            del line[k]
            line["dropped"] = dropped
    assert len(json.dumps(line)) <= LINE_CAP, "bench line over its size cap"
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=[None, "all", "train", "iou3d", "infer"])
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world, rank, local, info = setup_dist(args.gpus)
    from omni3d_amd import cpu_quota, respect_cpu_quota
    info["host_threads"] = {"torch_intra_op": respect_cpu_quota(), "cgroup_cpu_quota": cpu_quota(), "cpus_shown": os.cpu_count()}
    workload = args.workload or DEFAULT_WORKLOAD
    if workload == "iou3d":
        res = run_iou3d(args, world, rank)
    elif workload == "infer":
        from omni3d_amd.bench_train import run_infer
        res = run_infer(args, world, rank)
    else:
        from omni3d_amd.bench_train import run_train
        res = run_train(args, world, rank)
        if workload == "all":
            # second half of BASELINE.json's metric ("...; IoU3D boxes/sec", configs[4]) in the same driver-run line
            io = run_iou3d(args, world, rank)
            if rank == 0:
                res["iou3d"] = {k: io[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype",
                                                   "config", "roofline", "cpu_baseline")}
            if world == 1 and DEVICE == "cuda" and os.environ.get("OMNI_BENCH_SKIP_EXTRA") != "1":
                # the other two driver-visible numbers of the path -- inference on the same staged batch, and the
                # training step of configs[3]'s model (cubercnn_ResNet34_FPN, N = 1, 10 steps) -- short legs in the same line
                res["infer"] = extra_leg(["--workload", "infer", "--steps", "20", "--warmup", "3"], {})
                # the opt-in 6-term bf16-split form of the large-map point GEMMs (csrc/gemm_split.hip): its OWN number, never `value`
                res["train_split6"] = extra_leg(["--workload", "train", "--steps", "20", "--warmup", "3"],
                                                {"OMNI_GEMM_SPLIT": "6", "OMNI_BENCH_SKIP_CPU": "1", "OMNI_BENCH_SKIP_ROOFLINE": "1",
                                                 "OMNI_BENCH_SKIP_DROPIN": "1", "OMNI_BENCH_WINDOWS": "3", "OMNI_BENCH_CONDITION_STEPS": "50"})
                if isinstance(res["train_split6"], dict) and res["train_split6"].get("value"):
                    res["train_split6"]["operands"] = ("point GEMMs with >= 1024 Winograd tiles: fp32 operands split into 3 bf16 planes, 6 x "
                                                       "v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation (0.3x the fp32-MFMA kernel's error "
                                                       "vs float64); everything else as in the measured line")
                res["resnet34"] = extra_leg(["--workload", "train", "--steps", "10", "--warmup", "3"],
                                            {"OMNI_BENCH_CONFIG": "cubercnn_ResNet34_FPN.yaml", "OMNI_BENCH_SKIP_CPU": "1",
                                             "OMNI_BENCH_SKIP_ROOFLINE": "1", "OMNI_BENCH_SKIP_DROPIN": "1"})
    observe_ranks(world, local, info)
    if rank == 0:
        res["n_gpus"] = world
        res["launch"] = info
        if DEVICE != "cuda" or (world > 1 and not info.get("one_gpu_per_rank")):
            res["functional_check_only"] = ("ranks share a device or run host-compiled kernels: this line checks the N-rank code path, "
                                            "its numbers are not measurements of the product")
        print(json.dumps(compact_line(res, write_detail(res))))
    if world > 1:
        dist.barrier()      # rank 0 times the roofline kernels after the step loop: leave together
        dist.destroy_process_group()


DEFAULT_WORKLOAD = "all"

if __name__ == "__main__":
    main()
