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
import torch


class FeatureCut:
    """Splits backward at the FPN features: `cut(features)` hands detached copies to the heads (RPN, ROI heads), so
    `total.backward()` stops there with the heads' parameter gradients complete; `cut.backward()` then pushes the
    gradients that reached the copies through FPN + bottom-up.  Between the two the data-parallel step starts the
    all-reduce of the heads' gradient ranges (FlatSGD.all_reduce_begin("early"))."""

    def __init__(self):
        self.src, self.dst = None, None

    def __call__(self, features):
        self.src = features
        self.dst = {k: v.detach().requires_grad_(True) for k, v in features.items()}
        return self.dst

    def backward(self):
        pairs = [(self.src[k], self.dst[k].grad) for k in self.src if self.dst[k].grad is not None]
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
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            self.losses, self.total = self._phase_a()
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            self._phase_b()
        self.graph_a, self.graph_b = ga, gb

    def _phase_a(self):
        self.optimizer.zero_grad()
        losses = self.model(self.static_batch, self.static_packed)
        total = sum(losses.values())
        total.backward()
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
        # thread_local: RCCL's watchdog thread (N > 1) and the autograd worker may issue HIP calls while this thread
        # captures; only this thread's own unsafe calls should abort the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.losses, self.total = self._body()

    def _body(self):
        self.optimizer.zero_grad()
        losses = self.model(self.static_batch, self.static_packed)
        total = sum(losses.values())
        total.backward()
        return losses, total.detach()

    def __call__(self):
        """Replays zero_grad + forward + backward; gradients land in the optimizer's flat bucket.
        -> (loss dict, total) as static device tensors (overwritten by the next replay)."""
        self.graph.replay()
        return self.losses, self.total
