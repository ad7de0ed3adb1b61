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
