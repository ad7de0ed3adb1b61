"""hipGraph capture of the training step's device work.

The HIP training path has fixed shapes and no device->host synchronisation (fixed-size ROI blocks, packed
proposals + counts, sampling driven by device-side Exp(1) variates), so zero_grad + forward + the ten losses +
backward of one (batch shape, model) pair is a static sequence of ~800 kernel launches.  Replaying it as ONE
hipGraph removes the per-launch host cost (~17 ms of Python/ctypes enqueue per step, measured, against ~20 ms of
GPU work) and the inter-kernel launch gaps.  The learning-rate dependent part (gradient all-reduce, non-finite
scan, SGD update: 3-4 launches) stays eager so schedulers keep working on host floats.

The reference's loop is tools/train_net.py:do_train (:175-259); this object replaces its
`loss_dict = model(data); losses.backward()` pair for a pre-staged batch.  New input data is fed by copying
into the static tensors the graph was captured with (`static_batch` / `static_packed`)."""
import os

import torch

from ...functional import sum_vectors, total_loss


_units = {}


def _unit(t):
    """the gradient a scalar loss root starts from, built once per device: `t.backward()` without it launches a fill for
    ones_like(t) in every step"""
    key = (str(t.device), t.dtype)
    one = _units.get(key)
    if one is None:
        one = _units[key] = torch.ones((), dtype=t.dtype, device=t.device)
    return one


def _leaf_grad(t):
    """gradient of a cut's detached copy: .grad plus whatever its fan-in slot still withholds (functional.fanout)"""
    from ...functional import fanout_leftover
    left = fanout_leftover(t)
    if left is None:
        return t.grad
    return left if t.grad is None else t.grad + left


class lean_capture:
    """`torch.cuda.graph(g, pool=...)` without what its __enter__ adds in front of EVERY capture: torch.cuda.synchronize(), a full
    gc.collect() and torch.cuda.empty_cache().  A staged step is 14 captures; with one step captured per size bucket
    (solver/autoreplay.py) the empty_cache handed the eager path's cached blocks back to the driver 14 times per new bucket and the
    next eager iteration bought them again with hipMalloc (measured, round 6: -4 .. -15 GB of reserved memory per capture, single
    iterations of 0.4-2.7 s, profiles/r06_new_shape_*.txt).  The caller synchronises and collects ONCE before its first capture.
    OMNI_GRAPH_TORCH_CTX=1 restores torch's context manager."""
    _stream = None

    def __init__(self, graph, pool=None):
        self.graph, self.pool = graph, pool
        self.torch_ctx = torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local") if _TORCH_CTX else None

    def __enter__(self):
        if self.torch_ctx is not None:
            return self.torch_ctx.__enter__()
        if lean_capture._stream is None:
            lean_capture._stream = torch.cuda.Stream()
        self.stream = lean_capture._stream
        self.stream.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        kw = {"pool": self.pool} if self.pool is not None else {}
        self.graph.capture_begin(capture_error_mode="thread_local", **kw)

    def __exit__(self, *exc):
        if self.torch_ctx is not None:
            return self.torch_ctx.__exit__(*exc)
        self.graph.capture_end()
        self.ctx.__exit__(*exc)
        torch.cuda.current_stream().wait_stream(self.stream)
        return False


_TORCH_CTX = __import__("os").environ.get("OMNI_GRAPH_TORCH_CTX", "0") == "1"
_GRAPH_DUMP = __import__("os").environ.get("OMNI_GRAPH_DUMP", "")


class DeviceEvent:
    """A HIP event for DEVICE-side ordering only: hipEventDisableTiming | hipEventDisableSystemFence.  torch.cuda.Event (and
    Stream.wait_stream, which records one) creates its events with hipEventDisableTiming alone, and recording such an event ends with a
    system-scope release -- "cache writeback and invalidation, and the performance impact of those actions on the execution of
    following work" (hip_runtime_api.h).  The stage-end events of a replayed step are recorded on the critical-path stream after EVERY
    M_k while the weight-gradient stream keeps the L2s full of dirty lines; nobody on the host reads anything at those points.
    Recorded / waited on through the HIP runtime torch already has loaded; one object per use site, re-recorded every step."""
    _hip = None
    FLAGS = 0x2 | 0x20000000          # hipEventDisableTiming | hipEventDisableSystemFence

    def __init__(self):
        import ctypes
        if DeviceEvent._hip is None:
            DeviceEvent._hip = ctypes.CDLL("libamdhip64.so")
        self._ct = ctypes
        self.ev = ctypes.c_void_p()
        rc = DeviceEvent._hip.hipEventCreateWithFlags(ctypes.byref(self.ev), ctypes.c_uint(DeviceEvent.FLAGS))
        if rc != 0:
            raise RuntimeError("hipEventCreateWithFlags failed: %d" % rc)

    def record(self, stream):
        rc = DeviceEvent._hip.hipEventRecord(self.ev, self._ct.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("hipEventRecord failed: %d" % rc)

    def wait(self, stream):
        """`stream` waits for the work this event was last recorded behind"""
        rc = DeviceEvent._hip.hipStreamWaitEvent(self._ct.c_void_p(stream.cuda_stream), self.ev, self._ct.c_uint(0))
        if rc != 0:
            raise RuntimeError("hipStreamWaitEvent failed: %d" % rc)

    def __del__(self):
        try:
            if self.ev:
                DeviceEvent._hip.hipEventDestroy(self.ev)
        except Exception:
            pass


# MEASURED and left OFF (profiles/r06_ab_device_events.log): 10.80-10.81 ms with device-only events against 10.74-10.77 with torch's --
# the system-scope release is not what stretches the one stage boundary at which the weight-gradient queue is busy (~140 us instead of
# ~14, profiles/r06_ab_labels_cuts_gaps.log), and the fence-free markers are no cheaper for the command processor.
_PIPE_EVENTS = __import__("os").environ.get("OMNI_PIPE_EVENTS", "torch")      # "torch" | "device"


_W_CHUNKS = int(__import__("os").environ.get("OMNI_PIPE_W_CHUNKS", "1"))


class _GraphSeq:
    """several graphs replayed back to back (OMNI_PIPE_W_CHUNKS)"""

    def __init__(self):
        self.graphs = []

    def replay(self):
        for g in self.graphs:
            g.replay()

    def pool(self):
        return self.graphs[-1].pool()


def make_side_stream(device=None):
    """The weight-gradient stream.  The critical path runs on the main stream and is ~92 % busy (rocprofv3 trace, queue 1: 11.4 of
    12.4 ms); whatever the side stream runs beside it competes for the same CUs, and a PERSISTENT side kernel (the fc1-class weight
    gradient on the GEMM engine: one workgroup per CU for 0.57 ms) stops every main-stream kernel that cannot co-reside with it
    until it ends (a 50 us data gradient measured at 557 us).  Two knobs bound that:
      OMNI_SIDE_CUS=n       the side stream's queue is created on n of the 256 CUs (hipExtStreamCreateWithCUMask; the mask bits are
                            spread over the XCDs), so the critical path always finds 256 - n CUs free of weight-gradient work;
      OMNI_SIDE_PRIORITY=p  stream priority of the side stream (torch: lower value = higher priority; 0 = default).
    Defaults are the measured optimum (profiles/r03_ab_side_stream.log)."""
    import ctypes
    import os
    n = int(os.environ.get("OMNI_SIDE_CUS", str(SIDE_CUS_DEFAULT)))
    prio = int(os.environ.get("OMNI_SIDE_PRIORITY", "0"))
    if 0 < n < 256:
        hip = ctypes.CDLL("libamdhip64.so")
        dev = torch.cuda.current_device() if device is None else torch.device(device).index
        with torch.cuda.device(dev):
            words = (ctypes.c_uint32 * 8)()
            for bit in range(n):                # bits 0 .. n-1: KFD deals consecutive mask bits round-robin over the XCDs
                words[bit // 32] |= 1 << (bit % 32)
            st = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
            if rc == 0 and st.value:
                _MASKED_STREAMS.append(st)      # (never destroyed: lives as long as the process, like torch's pooled streams)
                return torch.cuda.ExternalStream(st.value, device=dev)
    if prio != 0:
        return torch.cuda.Stream(priority=prio)
    return torch.cuda.Stream()


SIDE_CUS_DEFAULT = 0            # 0 = no mask
_PIPE_ORDER = __import__("os").environ.get("OMNI_PIPE_ORDER", "interleaved")
# round 4: a cut at the pooled ROI features + the RPN losses deferred to that cut's stage, so stage 0 = forward + the FC heads'
# backward and W_0 (the fc1-class weight gradients, 1.1 ms of work) runs beside ROIAlign's / the RPN's backward instead of beside
# FPN + level 5 / 4 (A/B knob; the gradient bucket's stage layout follows it, solver/build.py)
POOL_CUT = __import__("os").environ.get("OMNI_PIPE_POOL_CUT", "1") != "0"
_PIPE_TIMING = __import__("os").environ.get("OMNI_PIPE_TIMING", "0") == "1"


def set_pipe_timing(flag):
    """bench.py turns the device timestamps on for every N > 1 run: the first multi-GPU curve should say WHERE a step spends its time
    (per stage: end of the critical-path graph M_k, of the weight-gradient graph W_k, of stage k's all-reduce X_k)"""
    global _PIPE_TIMING
    _PIPE_TIMING = bool(flag)


def pipe_timing_report(graphed, last=10):
    """OMNI_PIPE_TIMING=1: mean over the last steps of, per stage, when M_k / W_k ended on the device (ms after the step's first
    launch) and how long the host spent inside each graph launch (us) -- without a profiler attached"""
    torch.cuda.synchronize()
    recs = graphed._timing[-last:]
    if not recs:
        return ""
    n = len(recs[0]["m"])
    out = []
    for k in range(n):
        m = sum(r["t0"].elapsed_time(r["m"][k]) for r in recs) / len(recs)
        w = [r["t0"].elapsed_time(r["w"][k]) for r in recs if r["w"][k] is not None]
        x = [r["t0"].elapsed_time(r["x"][k]) for r in recs if r.get("x") and r["x"][k] is not None]
        out.append("M%d end %.3f ms%s%s" % (k, m, (", W%d end %.3f ms" % (k, sum(w) / len(w))) if w else "",
                                            (", X%d end %.3f ms" % (k, sum(x) / len(x))) if x else ""))
    host = {}
    for r in recs:
        for name, us in r["host"]:
            host.setdefault(name, []).append(us)
    out.append("host us per launch: " + ", ".join("%s %.0f" % (k, sum(v) / len(v)) for k, v in host.items()))
    return " | ".join(out)


def pipe_timing_table(graphed, last=10):
    """the same as numbers: {"M_end_ms": [...], "W_end_ms": [...], "X_end_ms": [...]} (None where a stage has no such part)"""
    torch.cuda.synchronize()
    recs = getattr(graphed, "_timing", [])[-last:]
    if not recs:
        return None

    def mean(key, k):
        v = [r["t0"].elapsed_time(r[key][k]) for r in recs if r.get(key) and r[key][k] is not None]
        return sum(v) / len(v) if v else None
    n = len(recs[0]["m"])
    return {"M_end_ms": [mean("m", k) for k in range(n)], "W_end_ms": [mean("w", k) for k in range(n)], "X_end_ms": [mean("x", k) for k in range(n)],
            "steps": len(recs), "note": "device timestamps after the step's first launch: end of stage k's critical-path graph (M), of its "
                                        "weight-gradient graph (W) and of the all-reduce calls issued behind it (X)"}
_MASKED_STREAMS = []


class FeatureCut:
    """Splits backward at the FPN features: `cut(features)` hands detached copies to the heads (RPN, ROI heads), so
    `total.backward()` stops there with the heads' parameter gradients complete; `cut.backward()` then pushes the
    gradients that reached the copies through FPN + bottom-up.  Between the two the data-parallel step starts the
    all-reduce of the heads' gradient ranges (FlatSGD.all_reduce_begin("early"))."""

    def __init__(self):
        self.src, self.dst = None, None

    def __call__(self, features):
        from ...functional import fanout
        self.src = features
        self.dst = {k: fanout(v.detach().requires_grad_(True)) for k, v in features.items()}
        return self.dst

    def backward(self):
        pairs = [(self.src[k], _leaf_grad(self.dst[k])) for k in self.src]
        pairs = [p for p in pairs if p[1] is not None]
        self.src, self.dst = None, None
        if pairs:
            torch.autograd.backward([p[0] for p in pairs], [p[1] for p in pairs])


class GraphedTwoPhase:
    """GraphedForwardBackward for data-parallel runs: graph A = zero_grad + forward + losses + backward of the heads,
    graph B = backward of FPN + bottom-up (same memory pool).  `__call__` replays A, starts the asynchronous all-reduce
    of the heads' gradients, replays B (RCCL runs beside the backbone's kernels), starts the all-reduce of the rest and
    returns the pending handles for FlatSGD.all_reduce_finish.  graphs=False runs the same sequence with eager launches
    (used when capture is refused, and by the CPU tests)."""

    def __init__(self, model, optimizer, batch, packed, warmup=3, graphs=True, group=None):
        self.model, self.optimizer, self.group = model, optimizer, group
        self.static_batch, self.static_packed = batch, packed
        self.cut = FeatureCut()
        model.feature_cut = self.cut
        self.graph_a = self.graph_b = None
        if not graphs:
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._phase_a()
                self._phase_b()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from ... import functional as HF
        from ...kernels import detmode
        detmode.prewarm(torch.cuda.current_device())
        prev_mode = HF.side_mode()
        HF.side_mode("inline")      # single-branch graphs: a captured fork / join would be replayed node by node from the host
        try:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                self.losses, self.total = self._phase_a()
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                self._phase_b()
        finally:
            HF.side_mode(prev_mode)
        self.graph_a, self.graph_b = ga, gb

    def _phase_a(self):
        self.optimizer.zero_grad()
        losses = self.model(self.static_batch, self.static_packed)
        total = total_loss(losses)             # == sum(losses.values()), two launches
        total.backward(_unit(total))
        return losses, total.detach()

    def _phase_b(self):
        self.cut.backward()

    def __call__(self):
        """-> (loss dict, total, pending all-reduce handles)"""
        if self.graph_a is not None:
            self.graph_a.replay()
        else:
            self.losses, self.total = self._phase_a()
        pending = self.optimizer.all_reduce_begin("early", self.group)
        if self.graph_b is not None:
            self.graph_b.replay()
        else:
            self._phase_b()
        pending += self.optimizer.all_reduce_begin("late", self.group)
        return self.losses, self.total, pending


class GraphedForwardBackward:
    def __init__(self, model, optimizer, batch, packed, warmup=3):
        assert torch.cuda.is_available(), "hipGraph capture needs the GPU"
        self.model, self.optimizer = model, optimizer
        self.static_batch, self.static_packed = batch, packed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up off the capture: allocator pools, lazily built constants
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        from ... import functional as HF
        from ...kernels import detmode
        detmode.prewarm(torch.cuda.current_device())
        prev_mode = HF.side_mode()
        HF.side_mode("inline")      # single-branch graph: a captured fork / join would be replayed node by node from the host
        # thread_local: RCCL's watchdog thread (N > 1) and the autograd worker may issue HIP calls while this thread
        # captures; only this thread's own unsafe calls should abort the capture
        try:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.losses, self.total = self._body()
        finally:
            HF.side_mode(prev_mode)

    def _body(self):
        self.optimizer.zero_grad()
        losses = self.model(self.static_batch, self.static_packed)
        total = total_loss(losses)             # == sum(losses.values()), two launches
        total.backward(_unit(total))
        return losses, total.detach()

    def __call__(self):
        """Replays zero_grad + forward + backward; gradients land in the optimizer's flat bucket.
        -> (loss dict, total) as static device tensors (overwritten by the next replay)."""
        self.graph.replay()
        return self.losses, self.total


class StageCuts:
    """Ordered cut points of one forward pass.  `cuts(x)` (a tensor or a dict of tensors) returns detached copies that the
    rest of the forward consumes; backward then runs in stages: `total.backward()` stops at the most recent cut, every
    `backward_last()` pushes the gradients that reached the newest remaining cut through the piece of the network in front of
    it, down to the cut before.  Gradients from several consumers of a copy accumulate in its `.grad` as usual."""

    def __init__(self):
        self.cuts = []

    def __len__(self):
        return len(self.cuts)

    def __call__(self, x):
        # (the copies are read by several consumers -- RPN head + ROIAlign, FPN lateral + the next DLA level --, possibly in
        # different backward stages: functional.fanout sums their gradients inside the consumers' kernels)
        from ...functional import fanout
        if isinstance(x, dict):
            dst = {k: fanout(v.detach().requires_grad_(True)) for k, v in x.items()}
        elif isinstance(x, (tuple, list)):
            dst = tuple(None if v is None else fanout(v.detach().requires_grad_(True)) for v in x)
        else:
            dst = fanout(x.detach().requires_grad_(True))
        self.cuts.append([x, dst, []])
        return dst

    def attach_root(self, loss):
        """a scalar whose backward is DEFERRED to the stage of the most recent cut (round 4: the RPN losses wait for the stage that
        also runs ROIAlign's backward, so that stage 0 ends -- and its weight-gradient graph starts -- right behind the FC heads)"""
        self.cuts[-1][2].append(loss)

    def reset(self):
        self.cuts.clear()

    def backward_last(self):
        src, dst, roots = self.cuts.pop()
        if isinstance(src, dict):
            pairs = [(src[k], _leaf_grad(dst[k])) for k in src]
        elif isinstance(src, (tuple, list)):
            pairs = [(a, _leaf_grad(b)) for a, b in zip(src, dst) if a is not None]
        else:
            pairs = [(src, _leaf_grad(dst))]
        pairs = [p for p in pairs if p[1] is not None]
        del src, dst
        if pairs or roots:
            torch.autograd.backward([p[0] for p in pairs] + list(roots), [p[1] for p in pairs] + [_unit(r) for r in roots])


class GraphedPipelined:
    """The training step as a software pipeline over two HIP streams.

    Backward is cut into stages -- heads | FPN + DLA levels 5, 4 | level 3 | level 2 .. stem by default (`stage_cut_at` of the
    bottom-up; two stages for a backbone without cut points).  A cut at level k is only valid together with the cuts at all
    lower levels: an uncut p_j feeds both its FPN lateral and level j+1, and would be back-propagated through twice.
    Stage k is recorded as TWO single-branch hipGraphs: M_k = everything on the critical path (data gradients, BatchNorm,
    ROIAlign, losses; M_0 also holds zero_grad + forward) and W_k = the stage's weight-gradient launches, which the backward
    functions queue instead of running (functional.side_mode("collect")).  Replay:

        main:  M_0 | M_1 | M_2 | ... | M_last | wait for the side stream
        side:        W_0 (after M_0) | W_1 (after M_1) | ... | W_last

    so weight gradients, which nothing on the critical path waits for, fill the CUs the small data-gradient kernels of the
    next stages leave idle.  Measured on MI355X (batch 4 x 512 x 512): 14.07 ms / step with 4 stages against 14.65 ms for the
    single graph (6 stages 14.18, 3 stages 14.42, 2 stages 14.92: when a weight-gradient graph starts, the critical-path
    kernels wait up to ~200 us for CU slots, which bounds the useful number of stages).  A single graph with parallel branches is
    no alternative: ROCm 7.2 replays such a graph node by node from the host, 12 ms per step instead of 0.7 ms; neither is a
    high-priority main stream (graph replays on it run 2x slower).

    Memory: M graphs share one pool, W graphs another, so a W graph's temporaries are never handed to a main-stream kernel.
    The INPUTS of the W graphs (saved activations, output gradients) live in the M pool and stay referenced for as long as
    the graphs exist, so no later M graph can be given their memory while a W graph may still be reading it (a few GB of the
    288 GB).  Nothing allocated by a warm-up step may be released inside a capture (observed on ROCm 7.2 / torch 2.10: the
    captured graph then replays garbage), hence the warm-up results are dropped before the first capture begins.
    With more than one rank the all-reduce of the heads' gradient ranges is issued behind W_0 on the side stream and overlaps
    the rest of backward; the remaining ranges follow W_last.  graphs=False runs the same stages eagerly (weight gradients
    inline), which is what the CPU / gloo tests exercise.
    `__call__` -> (loss dict, total, pending all-reduce handles)."""

    _warm_stream = None

    def __init__(self, model, optimizer, batch, packed, warmup=3, graphs=True, group=None, pools=None):
        from ... import functional as HF
        self.HF = HF
        self.model, self.optimizer, self.group = model, optimizer, group
        self.static_batch, self.static_packed = batch, packed
        # pools: (M pool, W pool) of an EARLIER captured step of the same process whose memory this one may share (round 6,
        # solver/autoreplay.py: one captured step per size bucket).  Two steps never run at the same time -- every step ends with the
        # main stream waiting for the weight-gradient stream -- and a step reads nothing it has not written itself in the same replay
        # except its static inputs and outputs (allocated outside / still referenced), so the buckets' activations can live in the
        # same memory: ~5 GB per bucket before, the largest bucket's footprint in total now.  `release_intermediates()` is what makes
        # a finished capture's memory available to the next one.
        self.pools = pools
        self.cuts = StageCuts()
        self._install()
        bottom_up = getattr(getattr(model, "backbone", None), "bottom_up", None)
        if bottom_up is not None and hasattr(type(bottom_up), "stage_cut"):
            import os
            if os.environ.get("OMNI_PIPE_CUTS") is not None:       # A/B knob: "" | "p2" | "p2,p3" | "p2,p3,p4" | "p2,p3,p4,p5"
                bottom_up.stage_cut_at = tuple(x for x in os.environ["OMNI_PIPE_CUTS"].split(",") if x)
        self.stages = None
        self._timing = []
        if not graphs:
            return
        assert torch.cuda.is_available(), "hipGraph capture needs the GPU"
        self.side = make_side_stream()
        prev_mode = HF.side_mode()
        HF.side_mode("inline")
        # ONE warm-up stream per process: torch's caching allocator binds a block to the stream it was allocated on, so a fresh stream per
        # captured step (one per size bucket under AutoReplay) left ~5 GB of cached blocks behind that no later stream could reuse -- the
        # reserved memory grew by that much per capture until a 0.2-2.5 s empty_cache() handed it back (round 6)
        if GraphedPipelined._warm_stream is None:
            GraphedPipelined._warm_stream = torch.cuda.Stream()
        warm = GraphedPipelined._warm_stream
        warm.wait_stream(torch.cuda.current_stream())
        # Warm-up off the capture: allocator pools, lazily built constants.  WITHOUT collectives: under the
        # reference's loader every rank decides for itself when its batch signature has repeated often enough to capture
        # (autoreplay.py), so a capture's warm-up steps on one rank would pair their all-reduces with another rank's real gradient
        # exchange -- a hang or silently mixed gradients.  The warm-up gradients are thrown away anyway; nothing in a capture
        # talks to another rank, and the replayed step's own exchange is the same (early, late) sequence as an eager step's.
        import time as _time
        _t0 = _time.perf_counter()
        muted = getattr(self.optimizer, "_exchange_muted", False)
        self.optimizer._exchange_muted = True
        try:
            with torch.cuda.stream(warm):
                for _ in range(warmup):
                    _, _, pending = self._eager()   # (results dropped here: nothing of a warm-up step may die inside a capture)
                    assert not pending
                    del pending
        finally:
            self.optimizer._exchange_muted = muted
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        _t1 = _time.perf_counter()
        HF.side_mode("collect")
        # forward branches (functional.set_branch_stream): the RPN's labelling + loss beside its proposal selection, inside M0.
        # MEASURED and left OFF (OMNI_PIPE_BRANCH=1 enables it): M0 ends 0.10 ms earlier on the device, but hipGraphLaunch of a graph
        # with a fork blocks the host for the length of a step on ROCm 7.2 (host time of the M0 launch 0.1 -> 11.9 ms), the later
        # stages are enqueued late and the step takes 12.5 ms instead of 11.4 (profiles/r04_ab_branch.log)
        self.branch = torch.cuda.Stream() if os.environ.get("OMNI_PIPE_BRANCH", "0") == "1" else None
        prev_branch = HF.set_branch_stream(self.branch)
        # deterministic split reductions (kernels/detmode.py): the critical-path graphs and the weight-gradient graphs replay
        # side by side, so each family gets its own block of arrival counters, allocated before the first capture starts
        from ...kernels import detmode, wino
        detmode.prewarm(torch.cuda.current_device())
        try:
            stages = []
            pool_m, pool_w = self.pools if self.pools is not None else (None, None)
            self._held = []                 # closures + their inputs: kept for the lifetime of the graphs (see class docstring)
            self.prologue = None
            while True:
                gm = torch.cuda.CUDAGraph()
                if not stages and self._split_forward(bottom_up):
                    gm = self._capture_stage0_split(bottom_up)
                    pool_m = gm.pool()
                elif not stages and self._split_labels():
                    gm, pool_w = self._capture_stage0_labels(pool_m, pool_w)
                    pool_m = gm.pool()
                else:
                    if _GRAPH_DUMP:
                        gm.enable_debug_mode()
                    with lean_capture(gm, pool_m), detmode.domain("M"):
                        if not stages:
                            self.losses, self.total = self._stage0()
                        else:
                            self.cuts.backward_last()
                    pool_m = gm.pool()
                    if _GRAPH_DUMP:                 # diagnostic (OMNI_GRAPH_DUMP=dir): the node list of every critical-path graph as a dot file
                        os.makedirs(_GRAPH_DUMP, exist_ok=True)
                        gm.debug_dump(os.path.join(_GRAPH_DUMP, "M%d.dot" % len(stages)))
                fns, keep = HF.side_take()
                # (measured, profiles/r04_ab_w_shift.log: carrying the heads' weight gradients into the NEXT stage's graph frees M1
                # -- 1.45 -> 0.79 ms, it is HBM-bound on the p2 maps and so is the fc1 weight gradient beside it -- and M2 pays it
                # back, 1.73 -> 2.45 ms: the two streams share one throughput, where the weight gradients land does not matter)
                gw = None
                if fns and _W_CHUNKS > 1:
                    # A/B (OMNI_PIPE_W_CHUNKS=n): the stage's weight gradients as n graphs replayed one after the other -- more graph
                    # boundaries on the weight-gradient queue, at which the command processor looks at the critical-path queue again
                    gw = _GraphSeq()
                    per = -(-len(fns) // _W_CHUNKS)
                    for c0 in range(0, len(fns), per):
                        g1 = torch.cuda.CUDAGraph()
                        with lean_capture(g1, pool_w), detmode.domain("W"), wino.batched_wgrads():
                            for fn in fns[c0:c0 + per]:
                                fn()
                        pool_w = g1.pool()
                        gw.graphs.append(g1)
                elif fns:
                    gw = torch.cuda.CUDAGraph()
                    with lean_capture(gw, pool_w), detmode.domain("W"), wino.batched_wgrads():
                        for fn in fns:          # (the Winograd-domain GEMMs of the stage leave together when the context closes)
                            fn()
                    pool_w = gw.pool()
                self._held.append((fns, keep))
                del fns, keep
                stages.append((gm, gw))
                if len(self.cuts) == 0:
                    break
            self.stages = stages
            self.pools = (pool_m, pool_w)
            self.phase_ms = {"warmup": 1e3 * (_t1 - _t0), "capture": 1e3 * (_time.perf_counter() - _t1)}      # (diagnostic: autoreplay's OMNI_AUTO_REPLAY_TIMING)
        finally:
            HF.set_branch_stream(prev_branch)
            HF.side_take()
            HF.side_mode(prev_mode)

    def release_intermediates(self):
        """Drop the references that pinned this step's intermediate tensors (the weight-gradient closures and their inputs) while its
        stages were being captured -- they had to outlive the capture of the LATER stages of the same step, whose graphs run beside the
        weight-gradient graphs that read them.  The graphs keep using the addresses; the blocks return to the graphs' private pools,
        where only a later capture INTO THE SAME POOLS (another size bucket's step, see `pools`) can be given them.  The loss tensors,
        logged scalars and static inputs stay referenced by their owners."""
        self._held = None

    # ---- round 4: the head of the forward pass ---------------------------------------------------------------------------------
    # The Winograd filter transforms of a pass (one launch, 260 MB moved, ~0.13 ms) sat at the very start of the critical path while
    # the weight-gradient stream idles through all of forward.  Stage 0 is therefore captured as THREE graphs: P (the transforms,
    # replayed on the side stream), M0a (zero_grad, preprocessing, stem, DLA level 0 / 1 -- no Winograd layer in there) and M0b
    # (the rest of stage 0), which starts behind both.  The forward is cut by a callable the bottom-up runs between level 1 and 2.
    # MEASURED and left OFF (OMNI_PIPE_PROLOGUE=1 enables it): 11.718 ms with and without.  The trace shows why: the head of M0a is
    # itself HBM-bound (the 191 MB zero-fill of the gradient bucket 25 -> 81 us, preprocessing 8 -> 31 us beside the transform's 260 MB),
    # so the first Winograd layer starts 7 us earlier, not 130 (profiles/r04_ab_prologue.log).
    @staticmethod
    def _split_forward(bottom_up):
        import os
        return (bottom_up is not None and hasattr(type(bottom_up), "fwd_split") and os.environ.get("OMNI_PIPE_PROLOGUE", "0") != "0")

    def _capture_stage0_split(self, bottom_up):
        from ...kernels import detmode
        HF = self.HF
        gp, ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(gp, capture_error_mode="thread_local"), detmode.domain("W"):
            pre = HF.wino_pretransform(self.model)
        self._held.append((pre,))
        state = {"done": False}

        def split():
            if not state["done"]:
                state["done"] = True
                ga.capture_end()
                gb.capture_begin(pool=pool, capture_error_mode="thread_local")
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        bottom_up.fwd_split = split
        try:
            with torch.cuda.stream(cap), detmode.domain("M"), HF.wino_preloaded(pre):
                ga.capture_begin(pool=pool, capture_error_mode="thread_local")
                self.losses, self.total = self._stage0()
                if not state["done"]:
                    raise RuntimeError("the forward pass never reached its split point")
                gb.capture_end()
        finally:
            bottom_up.fwd_split = None
        torch.cuda.current_stream().wait_stream(cap)
        self.prologue = (gp, ga)
        return gb

    # Round 6: the RPN's anchor labelling + sampling (rpn_match1 / rpn_match2 / top-k of the sampling keys / rpn_finalize: four
    # latency-bound launches, ~0.12 ms with the device otherwise idle) read the anchors and the ground truth only -- nothing the network
    # computes.  Stage 0 is captured as THREE graphs: L (those four launches; replayed on the weight-gradient stream, which idles through
    # all of forward), M0a (zero_grad .. RPN head) and M0b (losses, proposals, ROI heads, the heads' backward), which starts behind both.
    # Unlike the filter-transform prologue above, L moves no memory to speak of: it hides completely.  OMNI_PIPE_LABELS=0 switches it off.
    def _split_labels(self):
        import os
        rpn = getattr(self.model, "proposal_generator", None)
        return (os.environ.get("OMNI_PIPE_LABELS", "1") != "0" and rpn is not None and hasattr(rpn, "label_and_sample_anchors")
                and rpn.__dict__.get("_last_hw_list") is not None and getattr(rpn, "injected", None) is None and self.static_packed is not None)

    def _capture_stage0_labels(self, pool_m, pool_w):
        from ...kernels import detmode
        rpn = self.model.proposal_generator
        dev = next(self.model.parameters()).device
        anchors = rpn.anchor_generator.grid(rpn.__dict__["_last_hw_list"], dev)       # (cached by the warm-up pass)
        gl, ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with lean_capture(gl, pool_w), detmode.domain("W"):
            pre = rpn.label_and_sample_anchors(anchors, self.static_packed)
        self._held.append((pre, anchors))
        state = {"done": False}

        def split():
            if not state["done"]:
                state["done"] = True
                ga.capture_end()
                gb.capture_begin(pool=ga.pool(), capture_error_mode="thread_local")
            return pre
        if lean_capture._stream is None:
            lean_capture._stream = torch.cuda.Stream()
        cap = lean_capture._stream
        cap.wait_stream(torch.cuda.current_stream())
        rpn.__dict__["_label_split"] = split
        try:
            with torch.cuda.stream(cap), detmode.domain("M"):
                kw = {"pool": pool_m} if pool_m is not None else {}
                ga.capture_begin(capture_error_mode="thread_local", **kw)
                self.losses, self.total = self._stage0()
                if not state["done"]:
                    raise RuntimeError("the forward pass never asked for the anchor labels")
                gb.capture_end()
        finally:
            rpn.__dict__.pop("_label_split", None)
        torch.cuda.current_stream().wait_stream(cap)
        self.prologue = (gl, ga)
        return gb, gl.pool()

    def _install(self):
        """the model cuts its forward at THIS object's cut points (several captured steps may exist side by side -- one per size
        bucket, solver/autoreplay.py -- and eager iterations in between run uncut: `uninstall` after a capture / an eager staged step)"""
        self.model.feature_cut = self.cuts
        bottom_up = getattr(getattr(self.model, "backbone", None), "bottom_up", None)
        if bottom_up is not None and hasattr(type(bottom_up), "stage_cut"):
            bottom_up.stage_cut = self.cuts
        heads = getattr(self.model, "roi_heads", None)
        if heads is not None and hasattr(type(heads), "pool_cut") and POOL_CUT:
            heads.pool_cut = self._pool_cut

    def _pool_cut(self, xs):
        self._pool_cut_made = True
        return self.cuts(xs)
    _pool_cut_made = False

    def uninstall(self):
        heads = getattr(self.model, "roi_heads", None)
        if heads is not None and getattr(heads, "pool_cut", None) is not None and getattr(heads.pool_cut, "__self__", None) is self:
            heads.pool_cut = None
        if getattr(self.model, "feature_cut", None) is self.cuts:
            self.model.feature_cut = None
        bottom_up = getattr(getattr(self.model, "backbone", None), "bottom_up", None)
        if bottom_up is not None and getattr(bottom_up, "stage_cut", None) is self.cuts:
            bottom_up.stage_cut = None

    def _per_stage_exchange(self, n_graph_stages):
        """One exchange per backward stage needs the gradient bucket laid out for exactly these cut points (build_optimizer tagged the
        parameters from the bottom-up's `stage_cut_at`); anything else -- an A/B run with other cuts, a backbone without cut points --
        falls back to two phases: heads behind stage 0, the rest after the last stage.  Either way the SEQUENCE of all-reduce calls
        is the same (stage ranges in order, FlatOptimizer.exchange_chunks): ranks may mix the forms."""
        opt = self.optimizer
        bu = getattr(getattr(self.model, "backbone", None), "bottom_up", None)
        sig = tuple(getattr(bu, "stage_cut_at", ())) if bu is not None else ()
        if self._pool_cut_made:
            sig = sig + ("pool",)
        return (getattr(opt, "n_stages", 2) == n_graph_stages and getattr(opt, "stage_cut_signature", None) == sig
                and hasattr(opt, "stage_ranges"))

    def _late(self, per_stage):
        """what is left to exchange after the last stage: nothing when every stage issued its own ranges"""
        return [] if per_stage else self.optimizer.all_reduce_begin("late", self.group)

    def _stage0(self):
        self.cuts.reset()
        self.optimizer.zero_grad()
        losses = self.model(self.static_batch, self.static_packed)
        first, deferred, total = self._split_losses(losses)
        if deferred is None:
            total = total_loss(losses)
            total.backward(_unit(total))
            return losses, total.detach()
        # stage 0 back-propagates the ROI heads' losses only, down to the pooled ROI features; the RPN's two losses are the
        # roots of the next stage together with ROIAlign's backward.  (Both partial sums and the reported total come out of ONE
        # launch: functional.sum_vectors2.)
        self.cuts.attach_root(deferred)
        first.backward(_unit(first))
        return losses, total

    def _split_losses(self, losses):
        """-> (sum of the losses whose backward belongs to stage 0, sum of the deferred ones or None, their total as a plain value)"""
        heads = getattr(self.model, "roi_heads", None)
        if heads is None or getattr(getattr(heads, "pool_cut", None), "__self__", None) is not self or not len(self.cuts) or not self._pool_cut_made:
            return None, None, None
        vecs = getattr(losses, "vectors", None)
        if not vecs or sorted(n for _, names in vecs for n in names) != sorted(losses.keys()):
            return None, None, None
        late = [v for v, names in vecs if all(n.startswith("rpn/") for n in names)]
        early = [v for v, names in vecs if not all(n.startswith("rpn/") for n in names)]
        if not late or not early:
            return None, None, None
        from ...functional import sum_vectors2
        return sum_vectors2(early, late)

    def _eager(self):
        """all stages with eager launches (weight gradients wherever functional.side_mode() puts them)
        -> (loss dict, total, pending early all-reduce handles)"""
        losses, total = self._stage0()
        per_stage = self._per_stage_exchange(1 + len(self.cuts))
        pending = self.optimizer.all_reduce_begin(0 if per_stage else "early", self.group)
        k = 0
        while len(self.cuts):
            self.cuts.backward_last()
            k += 1
            if per_stage:
                self.HF.side_join()                  # (eager form: the stage's weight gradients are complete)
                pending += self.optimizer.all_reduce_begin(k, self.group)
        self.HF.side_join()
        self._eager_exchanged_all = per_stage
        return losses, total, pending

    def __call__(self):
        if self.stages is None:
            self._install()
            try:
                losses, total, pending = self._eager()
            finally:
                self.uninstall()
            return losses, total, pending + self._late(self._eager_exchanged_all)
        main, side = torch.cuda.current_stream(), self.side
        pending, n = [], len(self.stages)
        self._replay_per_stage = self._per_stage_exchange(n)
        ends = [None] * n
        timing = _PIPE_TIMING          # diagnostic (OMNI_PIPE_TIMING=1): device timestamps of every M_k / W_k end, see pipe_timing_report
        if timing:
            import time
            rec = {"t0": torch.cuda.Event(enable_timing=True), "m": [], "w": [None] * n, "x": [None] * n, "host": []}
            rec["t0"].record(main)
            self._timing.append(rec)
        dev_ev = _PIPE_EVENTS == "device" and not timing
        if dev_ev and getattr(self, "_dev_events", None) is None:
            self._dev_events = [DeviceEvent() for _ in range(n + 3)]            # stage ends, prologue fork / join, end of step

        def order(after, before, slot):
            """stream `after` waits for what `before` has been given so far (device-side ordering only)"""
            if dev_ev:
                e = self._dev_events[slot]
                e.record(before)
                e.wait(after)
            else:
                after.wait_stream(before)
        if self.prologue is not None:
            # P on the side stream (behind whatever the main stream did last: the optimizer's update), M0a on the main stream;
            # M0b -- stages[0] -- starts behind both
            gp, ga = self.prologue
            order(side, main, n)
            with torch.cuda.stream(side):
                gp.replay()
            ga.replay()
            order(main, side, n + 1)

        def launch_w(k):                              # W_k starts when M_k has finished ...
            gw = self.stages[k][1]
            if gw is None and k > 0 and not self._replay_per_stage:
                return []
            if dev_ev:
                ends[k].wait(side)
            else:
                side.wait_event(ends[k])
            with torch.cuda.stream(side):
                if gw is not None:
                    if timing:
                        h0 = time.perf_counter()
                    gw.replay()
                    if timing:
                        rec["host"].append(("W%d" % k, (time.perf_counter() - h0) * 1e6))
                        rec["w"][k] = torch.cuda.Event(enable_timing=True)
                        rec["w"][k].record(side)
                # stage k's gradients are final after W_k: their all-reduce rides behind it on the side stream and overlaps the
                # stages below (round 3: only the heads' ranges did, the backbone's 75 MB went out after the last stage)
                if self._replay_per_stage:
                    handles = self.optimizer.all_reduce_begin(k, self.group)
                else:
                    handles = self.optimizer.all_reduce_begin("early", self.group) if k == 0 else []
                if timing and handles:
                    # when did stage k's exchange finish?  A probe stream waits for the collectives (work.wait() is a stream-level
                    # wait for RCCL) and records -- neither the main nor the weight-gradient stream is held up by the measurement
                    if getattr(self, "_probe", None) is None:
                        self._probe = torch.cuda.Stream()
                    with torch.cuda.stream(self._probe):
                        for h in handles:
                            h[0].wait()
                        rec["x"][k] = torch.cuda.Event(enable_timing=True)
                        rec["x"][k].record(self._probe)
                return handles

        # host order M_0, M_1, W_0, M_2, W_1, ...: the next critical-path graph is always queued on the main stream before the
        # side-stream launch that depends on an event (measured: a graph launch behind a cross-stream event delays every
        # launch issued after it by ~150 us)
        if timing:
            for k in range(n):
                h0 = time.perf_counter()
                self.stages[k][0].replay()
                rec["host"].append(("M%d" % k, (time.perf_counter() - h0) * 1e6))
                ends[k] = torch.cuda.Event(enable_timing=True)
                ends[k].record(main)
                rec["m"].append(ends[k])
                if k >= 1:
                    pending += launch_w(k - 1)
            pending += launch_w(n - 1)
        elif _PIPE_ORDER == "mfirst":               # A/B: every critical-path graph first, then the weight-gradient graphs behind their events
            for k in range(n):
                self.stages[k][0].replay()
                ends[k] = self._dev_events[k] if dev_ev else torch.cuda.Event()
                ends[k].record(main)
            for k in range(n):
                pending += launch_w(k)
        else:
            for k in range(n):
                self.stages[k][0].replay()
                ends[k] = self._dev_events[k] if dev_ev else torch.cuda.Event()
                ends[k].record(main)
                if k >= 1:
                    pending += launch_w(k - 1)
            pending += launch_w(n - 1)
        order(main, side, n + 2)                      # ... and everything after the step waits for the last W
        return self.losses, self.total, pending + self._late(self._replay_per_stage)
