"""The drop-in loop at benchmark speed: staged hipGraph replay from INSIDE `model(data)` / `optimizer.step()`.

The reference's loop (tools/train_net.py:176-253) is

    loss_dict = model(data); losses = sum(loss_dict.values()); ...; optimizer.zero_grad(); losses.backward(); optimizer.step()

issued eagerly: ~900 launches per iteration, host-bound at ~22 ms on MI355X against 13.5 ms of device work.  The benchmarked
path replays the same launches as staged hipGraphs (cubercnn/solver/graphed.py:GraphedPipelined), but that object used to be
reachable only from bench.py.  `AutoReplay` closes the gap without touching the loop:

  * `build_optimizer(cfg, model)` attaches one to the model (`model._omni_auto`); `RCNN3D.forward` asks it first;
  * it watches the batch signature (number of images and their shapes).  After `warm` eager iterations with the same signature it
    captures the staged graphs on a static copy of the batch (the warm-up steps of the capture run on a snapshot of the BatchNorm
    running statistics, which is restored: a capture is not a training step);
  * from then on `model(data)` copies the new images / packed ground truth into the static tensors, replays ALL stages (forward,
    the ten losses, backward: the gradients are in the optimizer's flat bucket when it returns; with more than one rank the
    overlapped all-reduce of the heads' gradient ranges is already in flight) and returns a loss dict whose entries are backed by
    one autograd node;
  * `optimizer.zero_grad()` that follows leaves the bucket alone (it holds THIS iteration's gradients), `losses.backward()` runs
    that node, which has nothing left to compute and only records -- on the device, no host sync -- whether the upstream
    gradient was the all-ones of an unweighted `sum(loss_dict.values())`; `optimizer.step()` waits for the exchange and applies
    the fused update, which the device-side flag turns into a no-op if the loop did anything else with the losses (a weighted
    sum, a loss scale): the next `model(data)` then raises and names the remedy instead of training on wrong gradients;
  * a second `optimizer.zero_grad()` in the same iteration (the reference's "diverging: zero_grad, no step" branch,
    tools/train_net.py:245-247) clears the bucket for real;
  * eval mode runs eager launches.

Round 4: a cache of captured steps keyed by SIZE BUCKET.  Real Omni3D training resizes every image to a random short edge
(configs/Base.yaml:10-13: 25 values from 256 to 640; cubercnn/data/dataset_mapper.py:17-58), so the exact tuple of image shapes of a
batch almost never repeats -- but detectron2's ImageList pads every batch to multiples of 64 anyway, and everything behind the padded
tensor already takes the real image sizes as DEVICE data (`packed.image_hw`).  The signature is therefore (batch size, padded
height, padded width): the captured step holds its images in fixed-size slots of that shape, `model(data)` copies each new image
into the corner of its slot, and the preprocessing kernel masks the rest with the sizes of THIS batch (omni_preprocess_masked).  Up to
OMNI_AUTO_REPLAY_CACHE (16) captured steps are kept, least recently used evicted; a bucket is captured after `warm` eager iterations
of its own.  OMNI_AUTO_REPLAY=0 switches the mechanism off."""
import os
from collections import OrderedDict

import torch
from torch.autograd import Function

from ... import functional as HF
from ..modeling.targets import MAX_GT_PER_IMAGE, pack_targets

ENABLED = os.environ.get("OMNI_AUTO_REPLAY", "1") != "0"
# captured steps kept.  Round 6: 128 by default (was 16) AND bounded by memory -- a capture is refused room before it starts if the
# device has less than RESERVE_GB + 1.5 x the largest captured step free; least recently used buckets go first.  MI355X has 288 GB:
# the ~70 size buckets of the reference's loader (Base.yaml's 25 short edges x the datasets' aspect ratios on the 64 grid) fit, so
# the DEFAULT keeps the eager pass's padding grid and a replayed iteration is the same computation as an eager one.
CACHE = int(os.environ.get("OMNI_AUTO_REPLAY_CACHE", "128"))
RESERVE_GB = float(os.environ.get("OMNI_AUTO_REPLAY_RESERVE_GB", "24"))
# One pair of graph-private memory pools for ALL captured steps (round 6).  Measured before: 26 buckets = 130 GB of graph-private memory
# (5 GB each), the 288 GB device full after ~45 -- far fewer than the 82 buckets 320 batches of the reference-shaped stream touch.
# Steps never overlap in time and a step reads only what it wrote itself, so their activations can share addresses (graphed.py).
SHARE_POOLS = os.environ.get("OMNI_AUTO_REPLAY_SHARE_POOLS", "1") != "0"
# The shared pools start as two ARENAS: one big block reserved in each (critical-path pool, weight-gradient pool) before the first capture,
# so that every later capture carves its tensors out of memory the pools already own (freed neighbours coalesce inside a segment)
# instead of buying ~6 GB of new segments per bucket from the driver and handing them back in 0.2-0.9 s trims (round 6:
# 38.6 ms per iteration alone but 41-62 ms inside bench.py, where the allocator's state differed).  "0,0" switches the arenas off.
ARENA_GB = tuple(float(v) for v in os.environ.get("OMNI_AUTO_REPLAY_ARENA_GB", "20,6").split(","))
TRIM_GB = float(os.environ.get("OMNI_AUTO_REPLAY_TRIM_GB", "48"))        # reserved-but-unallocated memory above which a capture ends with empty_cache() ...
TRIM_FREE_GB = float(os.environ.get("OMNI_AUTO_REPLAY_TRIM_FREE_GB", "64"))   # ... but only once the DEVICE has less than this left: a trim is a 0.2-2 s
#   hipFree, and on a 288 GB device whose loop holds ~100 GB the slack is not needed by anybody (one of four bench.py runs of the final
#   head took 38.5 instead of 28.5 ms per iteration over the multi-scale region: ~1 s of stalls in its last quarter alone)
# A/B: stage new batches through pinned host buffers with stream-ordered copies.  MEASURED and left OFF: on this ROCm 7.2 host the 3 MB
# of image slots take ~20 ms to cross from pinned memory (37.1 against 11.9 ms per iteration, profiles/r04_dropin_phases.log); the pageable
# copies block the host until the previous step has drained, which costs 0.3 ms per iteration with the losses read every 1000th
PINNED = os.environ.get("OMNI_AUTO_REPLAY_PINNED", "0") == "1"
BUCKET = 64                                                           # ImageList's padding granularity (FPN size divisibility)
# Thrash guard (round 5): real Omni3D training draws 25 short edges x the datasets' aspect ratios -- 66 padded (height,
# width) pairs in 200 iterations of the synthetic Omni3D-shaped stream (bench.py `dropin_loop_multiscale_stream`), far more than the
# cache holds.  A capture costs several eager steps (three muted warm-up passes + seven captured stages), stalls the peer ranks that
# wait in their all-reduce, and an eager iteration on a never-seen shape is ~15x a replayed one: a cache that keeps evicting what it
# is about to need is worse than no cache, and a coarser grid that FITS the cache is far better than either -- the images sit in
# larger zero-padded slots (masked on the device with their real sizes), which the model treats like a batch whose largest image is
# that much larger.  Watched over a sliding window of GUARD_WINDOW training iterations; whenever the share of iterations that did
# not replay a cached bucket exceeds GUARD_MISS the guard moves one level up (at most once per window, one warning each):
#   level 0  buckets on the grid the eager pass pads to (64: replay and eager see the same tensor);
#   level k  (only with OMNI_AUTO_REPLAY_GRIDS set, see GRIDS) extents above GRID_ABOVE rounded up to GRIDS[k - 1].  The guard
#            does not climb one grid per window: it replays the extents of the last iterations under every candidate grid and takes
#            the FINEST one whose bucket count fits three quarters of the cache (measured on the 200-iteration stream: stepping
#            128 -> 256 one window at a time was still missing half the iterations when the run ended);
#   last     no NEW bucket is captured any more: cached buckets replay, everything else runs eager launches.
# An evicted bucket has to earn its `warm` eager iterations again (its counter is reset).
GUARD_WINDOW = int(os.environ.get("OMNI_AUTO_REPLAY_WINDOW", "48"))
GUARD_MISS = float(os.environ.get("OMNI_AUTO_REPLAY_MISS", "0.25"))
# Coarser bucket grids are OPT-IN since round 6: a slot padded to a 128 / 256 grid is NOT the tensor the eager pass
# or the reference builds (ImageList pads to size_divisibility = 64 of the batch maximum) -- BatchNorm's batch statistics then include
# the extra zero padding and the RPN samples among additional all-negative anchors, so replayed iterations would differ numerically
# from eager ones of the same run and from the reference.  OMNI_AUTO_REPLAY_GRIDS=128,256 trades that parity for a higher hit rate;
# without it the guard's only escalation is "no new captures".
GRIDS = tuple(int(v) for v in os.environ.get("OMNI_AUTO_REPLAY_GRIDS", "").split(",") if v)
GRID_ABOVE = int(os.environ.get("OMNI_AUTO_REPLAY_GRID_ABOVE", "256"))
ROW_FIELDS = ("gt", "gt_cls", "gt3d", "gtpose", "ign")           # (rows, ...) arrays indexed through gt_off / ign_off
FIXED_FIELDS = ("gt_off", "ign_off", "Ks", "v2r", "ratio", "image_hw")


class _Replayed(Function):
    """loss vector of a replayed step as an autograd node: backward has nothing to compute (the replay already put the
    gradients into the flat bucket) and only checks the upstream gradient on the device"""

    @staticmethod
    def forward(ctx, anchor, auto, *losses):
        ctx.auto, ctx.n = auto, len(losses)
        return torch.stack([t.detach().reshape(()) for t in losses])

    @staticmethod
    def backward(ctx, g):
        ctx.auto._on_backward(g)
        # the anchor's (zero) gradient: under the script's DDP wrapper it is the one parameter whose hook must fire (solver/ddp.py)
        return (torch.zeros_like(ctx.auto.anchor), None) + (None,) * ctx.n


class _Boundary(Function):
    """Loss dict of an EAGER iteration behind one autograd node that owns the iteration's real graph.  The training loop keeps
    its `loss_dict` / `losses` variables alive into the next `model(data)` call; a live graph keeps the parameters' AccumulateGrad
    nodes alive, and those are bound to the stream they were created on -- the default stream of an eager iteration.  A later
    hipGraph capture that meets such a node makes the autograd engine synchronise the capture stream with the default stream,
    which is illegal inside a capture (torch warns: "AccumulateGrad node's stream does not match ... may break CUDA graph capture";
    on ROCm 7.2 capture_end segfaults: tools/probes/capture_matrix.py, profiles/r03_capture_matrix.txt).  With the boundary the loop
    only ever holds this node; `release()` drops the inner graph before a capture, whatever the loop still references."""

    @staticmethod
    def forward(ctx, holder, anchor):
        # the inner tensors travel in `holder`, not as inputs: the engine must not see an edge into the inner graph
        ctx.holder = holder
        return tuple(t.detach() for t in holder["inner"])

    @staticmethod
    def backward(ctx, *grads):
        inner = ctx.holder.get("inner")
        if inner is None:
            raise RuntimeError("omni3d_amd: backward() through a loss dict of an earlier iteration (its graph was released when the "
                               "training step was captured)")
        ctx.holder["inner"] = None
        pairs = [(t, g) for t, g in zip(inner, grads) if g is not None and t.requires_grad]
        if pairs:
            torch.autograd.backward([p[0] for p in pairs], [p[1] for p in pairs])
        return None, torch.zeros_like(ctx.holder["anchor"])


class AutoReplay:
    def __init__(self, model, optimizer, warm=None, graphs=None):
        # eager iterations a size bucket runs before it is captured.  Round 6: 1 (was 2) -- with a cache that holds every bucket of the
        # reference's loader a capture is never wasted on a shape that will not come back, and every eager iteration saved is one replayed
        warm = int(os.environ.get("OMNI_AUTO_REPLAY_WARM", "1")) if warm is None else warm
        self.model, self.opt, self.warm = model, optimizer, warm
        self.graphs = graphs                     # None: hipGraphs on a GPU, eager staged launches elsewhere (CPU tests)
        self.cache = OrderedDict()               # bucket -> captured step (stepper, static batch / targets, logged scalars)
        self.counts = {}                         # bucket -> iterations seen
        self.failed = None
        self.anchor = None
        self.bad = None                          # device float: != 0 when backward saw a non-unit upstream gradient
        self.bad_host, self.bad_event, self.bad_armed = None, None, False
        self.replays = 0
        self.captures = 0
        self.evictions = 0
        self.recaptures = 0                      # captures of a bucket that had been captured (and evicted) before
        self.eager_iters = 0                     # training iterations this object sent down the eager path
        self.ever = set()                        # buckets captured at least once
        self.level = 0                           # thrash-guard level (see GUARD_WINDOW)
        self.window = []                         # last GUARD_WINDOW iterations: True = replay of a cached bucket, False = miss
        self.level_at = 0                        # iteration count at the last escalation
        self.iters = 0
        self.granularity = self._model_bucket(model)
        self.extents = []                        # (batch size, max height, max width) of the recent iterations (see _note)
        self.evictions_at_level = 0
        self.holders = []                        # inner graphs of the eager iterations' loss dicts (see _Boundary)
        self.pools = None                        # (M pool, W pool) shared by the captured steps of every bucket (SHARE_POOLS)
        self._eager_anchor = None
        self.busy = False                        # True while a capture drives the model itself
        optimizer._auto = self

    # ---- signature / state machine -----------------------------------------------------------------------------------
    @staticmethod
    def _model_bucket(model):
        """padding granularity of the eager pass: `backbone.size_divisibility` (ImageList.from_tensors); a replayed pass must see the
        same padded tensor as an eager one, so the bucket grid is that value (64 for every FPN-with-p6 configuration of the reference)"""
        div = int(getattr(getattr(model, "backbone", None), "size_divisibility", 0) or 0)
        return div if div > 0 else BUCKET

    def signature(self, batch, record=True):
        """(batch size, padded height, padded width): what every shape behind ImageList.from_tensors depends on"""
        H = max(b["image"].shape[-2] for b in batch)
        W = max(b["image"].shape[-1] for b in batch)
        if not record:
            return self._bucket(len(batch), H, W, self.level)

        self.extents.append((len(batch), H, W))
        if len(self.extents) > 4 * GUARD_WINDOW:
            del self.extents[: len(self.extents) - 4 * GUARD_WINDOW]
        return self._bucket(len(batch), H, W, self.level)

    def _bucket(self, B, H, W, level):
        g = self.granularity
        coarse = GRIDS[min(level, len(GRIDS)) - 1] if (level >= 1 and GRIDS) else 0

        def up(v):
            q = -(-v // g) * g
            if coarse and q > GRID_ABOVE and coarse % g == 0:
                q = -(-v // coarse) * coarse
            return q
        return (B, up(H), up(W))

    def stats(self):
        """what the cache did so far (bench.py `dropin_loop_multiscale`, tests)"""
        n = max(self.iters, 1)
        return {"iterations": self.iters, "replays": self.replays, "eager": self.eager_iters, "captures": self.captures, "recaptures": self.recaptures,
                "evictions": self.evictions, "buckets_seen": len(self.counts) + len([k for k in self.ever if k not in self.counts]),
                "buckets_cached": len(self.cache), "hit_rate": self.replays / n, "guard_level": self.level, "bucket_granularity": self.granularity,
                "cached_gb": round(sum(e.get("bytes", 0) for e in self.cache.values()) / (1 << 30), 2)}

    def _note(self, hit):
        """sliding-window bookkeeping of the thrash guard; escalates at most once per window"""
        self.iters += 1
        self.window.append(bool(hit))
        if len(self.window) > GUARD_WINDOW:
            self.window.pop(0)
        if self.level > len(GRIDS) or len(self.window) < GUARD_WINDOW or self.iters - self.level_at < GUARD_WINDOW:
            return
        if len(self.cache) < max(CACHE, 1) and self.evictions == self.evictions_at_level:
            return                                # still filling the cache (for the first time on this grid): misses are warm-up, not thrash
        miss = 1.0 - sum(self.window) / len(self.window)
        if miss <= GUARD_MISS:
            return
        import warnings
        seen = len(self.counts)
        # the finest grid above the current one under which the recent extents fit 3/4 of the cache; none does: the coarsest, then freeze
        nxt = self.level + 1
        while nxt < len(GRIDS) and len({self._bucket(*e, nxt) for e in self.extents}) > 0.75 * max(CACHE, 1):
            nxt += 1
        self.level = nxt
        self.level_at = self.iters
        if self.level <= len(GRIDS):
            # the buckets of the finer grid will not be asked for again: release their graphs, and let the new grid fill the cache like
            # a fresh start (misses while it fills are warm-up, not thrash: see the test above)
            self.cache.clear()
            self.pools = None
            self.counts.clear()
            self.window.clear()
            self.evictions_at_level = self.evictions
        if self.level <= len(GRIDS):
            warnings.warn(f"omni3d_amd: {miss:.0%} of the last {GUARD_WINDOW} iterations missed the {CACHE}-entry cache of captured training "
                          f"steps ({seen} size buckets seen): rounding extents above {GRID_ABOVE} up to {GRIDS[self.level - 1]} from now on")
        else:
            warnings.warn(f"omni3d_amd: the captured-step cache still misses {miss:.0%} of the iterations; no new size bucket is captured "
                          "any more (cached buckets replay, the rest runs eager launches).  OMNI_AUTO_REPLAY_CACHE raises the cache size")

    # (kept for callers / tests that look at the most recently used captured step)
    @property
    def stepper(self):
        return next(reversed(self.cache.values()))["stepper"] if self.cache else None

    def forward(self, batched_inputs):
        """-> loss dict of a replayed step, or None (the caller runs the eager path)"""
        if not ENABLED or self.failed is not None or not self.model.training:
            return None
        self._raise_if_poisoned()
        sig = self.signature(batched_inputs)
        entry = self.cache.get(sig)
        level = self.level
        self._note(entry is not None)
        if self.level != level:                  # the guard changed the grid (and emptied the cache): this batch's bucket on the new one
            sig = self.signature(batched_inputs, record=False)
            entry = self.cache.get(sig)
        self.counts[sig] = self.counts.get(sig, 0) + 1
        if entry is None:
            if self.counts[sig] <= self.warm or self.level > len(GRIDS):
                self.eager_iters += 1
                return None
            try:
                self._make_room()
                entry = self._capture(batched_inputs, sig)
            except Exception as e:  # noqa: BLE001 -- capture refused: stay on eager launches, say why once
                self._drop()
                self.failed = f"{type(e).__name__}: {str(e)[:200]}"
                import warnings
                warnings.warn(f"omni3d_amd: staged-graph capture of the training step failed ({self.failed}); running eager launches")
                return None
            if sig in self.ever:
                self.recaptures += 1
            self.ever.add(sig)
            self.cache[sig] = entry
            while len(self.cache) > max(CACHE, 1):
                old_sig, _ = self.cache.popitem(last=False)   # least recently used: its graphs and their private pools are released
                self.counts.pop(old_sig, None)                # ... and it has to earn its warm-up iterations again
                self.evictions += 1
        self.cache.move_to_end(sig)
        if getattr(self.opt, "_replay_state", None) is not None:
            # the previous replayed gradients were neither applied (step) nor dropped (second zero_grad): a loop that accumulates
            # over several forward passes.  The replay zeroes the bucket itself, so that pattern needs eager launches.
            self._drop()
            self.opt._replay_state = None
            self.failed = "forward called again before optimizer.step()"
            import warnings
            warnings.warn("omni3d_amd: gradient accumulation over several model(data) calls detected; staged-graph replay switched off")
            return None
        self._stage(entry, batched_inputs)
        losses, total, pending = entry["stepper"]()
        self.opt._replay_state = {"pending": pending, "zero_grads_seen": 0}
        self.replays += 1
        from ..modeling.layers import PARAM_EPOCH
        PARAM_EPOCH[0] += 1      # the replay moved the BatchNorm running statistics through raw pointers: eval-mode caches (layers.py, InferReplay)
                                 # are stale even if this iteration's update is dropped
        names = list(losses.keys())
        out = _Replayed.apply(self.anchor, self, *[losses[k].detach() for k in names])     # detached: nothing upstream of the node
        from ...d2.events import get_event_storage, has_event_storage
        if has_event_storage():                      # the logged scalars live in static tensors the replay refreshed
            for m, saved in entry["logs"]:
                m.pending_logs = dict(saved)
            self.model.flush_logs(get_event_storage())
        return HF.LossDict({k: out[i] for i, k in enumerate(names)})

    def wrap_eager(self, losses):
        """eager iteration: hand the loop a loss dict whose real graph this object can release (see _Boundary)"""
        if not ENABLED or self.failed is not None or not losses:
            return losses
        names = list(losses.keys())
        if not any(losses[k].requires_grad for k in names):
            return losses
        anchor = getattr(self.model, "_omni_ddp_anchor", None)
        if anchor is None:
            if self._eager_anchor is None:
                self._eager_anchor = torch.zeros(1, device=losses[names[0]].device, requires_grad=True)
            anchor = self._eager_anchor
        holder = {"inner": tuple(losses[k] for k in names), "anchor": anchor}
        self.holders = [h for h in self.holders if h.get("inner") is not None][-4:] + [holder]
        out = _Boundary.apply(holder, anchor)
        wrapped = HF.LossDict({k: o for k, o in zip(names, out)})
        return wrapped

    def release_eager_graphs(self):
        import gc
        for h in self.holders:
            h["inner"] = None
        self.holders = []
        # The graphs die by reference count when the holders let go; the collector is the safety net for a cycle through one of them.
        # Only the young generations: the holders are at most four iterations old, and a FULL collection walks the whole heap of the
        # training process (dataset dicts, loader state) -- measured 83 ms of a 210 ms capture (round 6, OMNI_AUTO_REPLAY_TIMING=1).
        gc.collect() if os.environ.get("OMNI_AUTO_REPLAY_FULL_GC") == "1" else gc.collect(1)

    def _drop(self):
        """forget every captured step (a failure, or a loop the protocol does not cover)"""
        self.cache.clear()
        self.pools = None
        self._arena_graphs = None
        self.counts.clear()
        self.model.feature_cut = None
        bu = getattr(getattr(self.model, "backbone", None), "bottom_up", None)
        if bu is not None and hasattr(bu, "stage_cut"):
            bu.stage_cut = None

    # ---- capture -----------------------------------------------------------------------------------------------------
    def _free_bytes(self):
        """device memory a new capture's private pools can still get from the driver (blocks cached in OTHER graphs' pools are not
        available to it; the general pool's cached blocks are handed back first if the figure is short)"""
        return torch.cuda.mem_get_info()[0]

    def _make_room(self):
        """memory bound of the cache: least recently used captured steps are released until RESERVE_GB + 1.5 x the largest captured
        step so far is free (their graphs' private pools go back to the allocator)"""
        if self.model.device.type != "cuda" or not self.cache:
            return
        need = RESERVE_GB * (1 << 30) + 1.5 * max((e.get("bytes", 0) for e in self.cache.values()), default=0)
        if self._free_bytes() < need:
            torch.cuda.empty_cache()
        while self.cache and self._free_bytes() < need:
            old_sig, old = self.cache.popitem(last=False)
            self.counts.pop(old_sig, None)
            self.evictions += 1
            del old
            if not self.cache:
                self.pools = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()

    def _make_arenas(self, dev):
        """two graph-private pools, each holding one big FREE block: a dummy capture allocates it inside the pool and lets it go again"""
        from .graphed import lean_capture
        pools, keep = [], []
        torch.cuda.synchronize()
        for gb in ARENA_GB[:2]:
            pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            free = torch.cuda.mem_get_info()[0]
            n = int(min(gb * (1 << 30), max(free - RESERVE_GB * (1 << 30), 0)))
            with lean_capture(g, pool):
                t = torch.empty(max(n, 1 << 20), dtype=torch.uint8, device=dev)
                t[:256].zero_()                     # (a capture must hold a node)
                del t
            pools.append(pool)
            keep.append(g)                          # the pool lives as long as a graph captured into it does
        self.pools = tuple(pools)
        self._arena_graphs = keep

    def _capture(self, batch, sig):
        from .graphed import GraphedPipelined
        model, dev = self.model, self.model.device
        B, Hb, Wb = sig
        # the images' slots: every later batch of this bucket is copied into their corners
        slots = torch.zeros((B, 3, Hb, Wb), dtype=batch[0]["image"].dtype, device=dev)
        sb = []
        for n, b in enumerate(batch):
            c = {k: v for k, v in b.items() if k not in ("image", "instances")}
            h, w = b["image"].shape[-2:]
            slots[n, :, :h, :w].copy_(b["image"], non_blocking=True)
            c["image"] = slots[n]
            if "instances" in b:
                c["instances"] = b["instances"]
            sb.append(c)
        import time as _time
        _t = [_time.perf_counter()]
        self.release_eager_graphs()                      # no live eager graph (and its default-stream AccumulateGrad nodes) during capture
        _t.append(_time.perf_counter())
        packed = model.prepack(batch)                    # (real image sizes: packed.image_hw)
        cap = B * int(os.environ.get("OMNI_AUTO_REPLAY_ROWS", MAX_GT_PER_IMAGE))
        for f in ROW_FIELDS:                              # fixed capacity: any later batch's rows fit
            t = getattr(packed, f)
            pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[: t.shape[0]] = t
            setattr(packed, f, pad)
        packed.slotted = True                            # RCNN3D.preprocess_image masks the slots with packed.image_hw on the device
        graphs = self.graphs if self.graphs is not None else (dev.type == "cuda")
        mem0 = torch.cuda.memory_reserved() if dev.type == "cuda" else 0
        if graphs and SHARE_POOLS and self.pools is None and dev.type == "cuda" and max(ARENA_GB) > 0:
            self._make_arenas(dev)
        # a capture is not a training step: its warm-up passes must not move the BatchNorm running statistics
        bufs = [(b, b.detach().clone()) for b in model.buffers()]
        self.busy = True
        try:
            # (one muted warm-up pass inside the capture instead of three: the bucket's eager iterations have already built every
            # lazily-made constant of this shape; OMNI_AUTO_REPLAY_CAPTURE_WARMUP restores more)
            stepper = GraphedPipelined(model, self.opt, sb, packed, graphs=graphs, warmup=int(os.environ.get("OMNI_AUTO_REPLAY_CAPTURE_WARMUP", "1")),
                                       pools=self.pools if SHARE_POOLS else None)
        finally:
            self.busy = False
            with torch.no_grad():
                for b, saved in bufs:
                    b.copy_(saved)
        _t.append(_time.perf_counter())
        if os.environ.get("OMNI_AUTO_REPLAY_TIMING") == "1":
            print("capture %s: release+gc %.1f ms, stage+capture %.1f ms (%s)" % (sig, 1e3 * (_t[1] - _t[0]), 1e3 * (_t[2] - _t[1]),
                  ", ".join("%s %.1f" % kv for kv in getattr(stepper, "phase_ms", {}).items())), flush=True)
        if stepper.stages is not None:
            stepper.uninstall()                          # replays need no cut points; eager iterations of other buckets run uncut
            if SHARE_POOLS:
                # every bucket's graphs allocate from ONE pair of private pools (graphed.py `pools`): this step's intermediates are
                # released to them now, so the next bucket's capture reuses that memory instead of reserving its own ~5 GB
                stepper.release_intermediates()
                self.pools = stepper.pools
                # segments of the shared pools that this capture left completely unused go back to the driver -- ONCE per captured step
                # (torch's own capture context did it in front of each of the 14 stage captures; without it at all the pools'
                # reserved memory grew to 222 GB over 47 buckets of different tensor sizes: profiles/r06_new_shape_*.txt)
                # -- and only when the slack is worth a trip to the driver: hipFree of several GB takes 0.2-2 s (measured per capture)
                if (torch.cuda.memory_reserved() - torch.cuda.memory_allocated() > TRIM_GB * (1 << 30)
                        and torch.cuda.mem_get_info()[0] < TRIM_FREE_GB * (1 << 30)):
                    torch.cuda.empty_cache()
        if self.anchor is None:
            self.anchor = getattr(model, "_omni_ddp_anchor", None)
            if self.anchor is None:
                self.anchor = torch.zeros(1, device=dev, requires_grad=True)
            self.bad = torch.zeros(1, dtype=torch.float32, device=dev)
        self.captures += 1
        # what the components would log this iteration (static tensors of the captured pass; flush_logs pops them)
        logs = [(m, dict(m.pending_logs)) for m in (model.proposal_generator, model.roi_heads) if hasattr(m, "pending_logs")]
        return {"stepper": stepper, "batch": sb, "packed": packed, "slots": slots, "logs": logs,
                "bytes": max(0, torch.cuda.memory_reserved() - mem0) if dev.type == "cuda" else 0}

    def _stage(self, entry, batch):
        """new data into the tensors the graphs were captured on.  With OMNI_AUTO_REPLAY_PINNED=1 everything crosses PCIe from pinned
        staging buffers (two sets per captured step, taken in turn, reused only after the copies that read a set have executed) with
        stream-ordered copies, so that the host does not wait for the device here; see PINNED for why that is not the default."""
        dev = self.model.device
        sizes = [(b["image"].shape[-2], b["image"].shape[-1]) for b in batch]
        new = pack_targets(batch, sizes, getattr(self.model.roi_heads, "virtual_focal", 512.0), with_gt=True)
        sp = entry["packed"]
        for f in ROW_FIELDS:
            if getattr(new, f).shape[0] > getattr(sp, f).shape[0]:
                raise ValueError(f"{getattr(new, f).shape[0]} rows of {f} exceed the static capacity {getattr(sp, f).shape[0]}")
        for s_, b in zip(entry["batch"], batch):
            for k in ("K", "height", "width"):
                if k in b:
                    s_[k] = b[k]
        pin = None
        if dev.type == "cuda" and PINNED:
            sets = entry.setdefault("pinned", [])
            turn = entry["turn"] = (entry.get("turn", -1) + 1) % 2
            if len(sets) <= turn:
                st = {"img": torch.zeros(entry["slots"].shape, dtype=entry["slots"].dtype).pin_memory(), "event": None}
                for f in ROW_FIELDS + FIXED_FIELDS:
                    t = getattr(sp, f)
                    st[f] = torch.zeros(t.shape, dtype=t.dtype).pin_memory()
                sets.append(st)
            pin = sets[turn]
            if pin["event"] is not None:
                pin["event"].synchronize()                      # its copies of two iterations ago have executed (normally long ago)
        host_imgs = pin is not None and all(not b["image"].is_cuda for b in batch)
        for n, b in enumerate(batch):
            h, w = sizes[n]
            if host_imgs:
                pin["img"][n, :, :h, :w].copy_(b["image"])                            # host -> pinned (the rest of the slot is masked by image_hw)
            else:
                entry["slots"][n, :, :h, :w].copy_(b["image"], non_blocking=True)
        if host_imgs:
            entry["slots"].copy_(pin["img"], non_blocking=True)                       # ONE contiguous, asynchronous copy of all slots
        for f in ROW_FIELDS + FIXED_FIELDS:
            src, dst = getattr(new, f), getattr(sp, f)
            n = src.shape[0] if f in ROW_FIELDS else None
            if pin is not None:
                if n is None:
                    pin[f].copy_(src)
                    dst.copy_(pin[f], non_blocking=True)
                elif n > 0:
                    pin[f][:n].copy_(src)
                    dst[:n].copy_(pin[f][:n], non_blocking=True)
            elif n is None:
                dst.copy_(src.to(dev, non_blocking=True), non_blocking=True)
            else:
                dst[:n].copy_(src.to(dev, non_blocking=True), non_blocking=True)
        if pin is not None:
            pin["event"] = torch.cuda.Event()
            pin["event"].record()
        sp.num_gt, sp.num_ign = new.num_gt, new.num_ign

    # ---- backward / step protocol --------------------------------------------------------------------------------------
    def _on_backward(self, g):
        """upstream gradient of the replayed loss vector: all ones for `sum(loss_dict.values()).backward()`.  Anything else would
        need other gradients than the ones the replay produced: flag it on the device (the fused update skips), tell the host
        at the next iteration."""
        with torch.no_grad():
            self.bad.copy_((g != 1.0).any().to(torch.float32).reshape(1))

    def arm_check(self):
        """called by optimizer.step(): this step consumed replayed gradients; read the flag back asynchronously"""
        if self.bad is None:
            return
        if self.bad.is_cuda:
            if self.bad_host is None:
                self.bad_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self.bad_host.copy_(self.bad, non_blocking=True)
            self.bad_event = torch.cuda.Event()
            self.bad_event.record()
        else:
            self.bad_host = self.bad.clone()
            self.bad_event = None
        self.bad_armed = True

    def _raise_if_poisoned(self):
        if not self.bad_armed:
            return
        if self.bad_event is not None and not self.bad_event.query():
            return                                # not there yet: look again at the next iteration
        self.bad_armed = False
        if float(self.bad_host[0]) != 0.0:
            self._drop()
            self.failed = "non-unit upstream gradient"
            raise RuntimeError(
                "omni3d_amd: the training loop back-propagated something other than the unweighted sum(loss_dict.values()) through a "
                "replayed step (a loss weight or scale applied outside the model).  That iteration's update was skipped on the device. "
                "Set OMNI_AUTO_REPLAY=0 (eager launches honour any upstream gradient), or fold the weights into cfg.MODEL.*.LOSS_W_*.")


def attach(model, optimizer):
    """build_optimizer hook: one AutoReplay per (model, optimizer) pair; RCNN3D.forward finds it as `model._omni_auto`"""
    inner = model.module if hasattr(model, "module") else model
    if not ENABLED or not hasattr(inner, "prepack"):
        return None
    auto = AutoReplay(inner, optimizer)
    inner.__dict__["_omni_auto"] = auto
    return auto
