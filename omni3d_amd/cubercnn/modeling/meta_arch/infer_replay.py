"""`model(batch)` in eval mode from a captured hipGraph (round 4).

RCNN3D.inference (rcnn3d.py:79-112 of the reference) is ~570 launches of fixed shape for a given padded batch size -- backbone, RPN,
box head, batched per-class NMS, cube head, fused decode -- followed by one device->host copy of the detection counts.  Issued
eagerly the pass is bound by the host (10.5 ms per batch of four 512 x 512 images against 7 ms of kernel time).  Like the training
step (solver/autoreplay.py) the device half is captured once per size bucket -- (batch size, padded height, padded width); the real
image sizes are device data (`packed.image_hw`) -- and replayed: `model(batch)` copies the images into the corners of their slots,
refreshes the intrinsics / ratios / sizes in the static `packed`, launches ONE graph and runs the host half (counts, Instances,
postprocess) on its output tensors.  Results are the eager path's: the same kernels on the same values.

Oracle-2D inputs, a caller-supplied `packed`, CPU tensors and OMNI_INFER_REPLAY=0 take the eager path; so does the first pass
of every bucket (it also fills the per-shape caches -- anchors, batch indices -- a capture must not create)."""
import os
from collections import OrderedDict

import torch

from ..targets import pack_targets

ENABLED = os.environ.get("OMNI_INFER_REPLAY", "1") != "0"
CACHE = int(os.environ.get("OMNI_INFER_REPLAY_CACHE", "8"))
FIELDS = ("Ks", "v2r", "ratio", "image_hw")          # what the inference pass reads of the packed batch description


class InferReplay:
    def __init__(self, model, warm=1, graphs=None):
        self.model, self.warm = model, warm
        self.graphs = graphs                 # None: hipGraphs on a GPU, nothing elsewhere; False: the same staging with eager launches (CPU tests)
        self.cache, self.counts = OrderedDict(), {}
        self.failed = None
        self.replays = self.captures = 0
        self.digest, self._tensors, self._side = None, None, None

    def run(self, batched_inputs, context):
        """-> (raw device outputs, image sizes) of a replayed pass, or None (the caller runs the eager pass).  `context`: a callable
        returning the context manager the eager pass runs its device half under (Winograd tile choice, filter-transform scope)"""
        if not ENABLED or self.failed is not None or not batched_inputs:
            return None
        img0 = batched_inputs[0]["image"]
        if (self.model.device.type != "cuda" and self.graphs is not False) or any("oracle2D" in b for b in batched_inputs):
            return None
        # A captured pass reads the weights through their pointers (live), but values DERIVED from them outside the graph -- the
        # cached eval-mode BatchNorm coefficients -- are baked in as the tensors they were at capture time: any parameter / buffer
        # write since then (an optimizer step, load_state_dict) drops the captured passes
        digest = self._digest()
        if digest != self.digest:
            self.cache.clear()
            self.counts.clear()
            self.digest = digest
        sig = self.signature(batched_inputs) + (str(img0.dtype),)
        self.counts[sig] = self.counts.get(sig, 0) + 1
        entry = self.cache.get(sig)
        if entry is None:
            if self.counts[sig] <= self.warm:
                return None
            try:
                entry = self._capture(batched_inputs, sig, context)
            except Exception as e:  # noqa: BLE001 -- capture refused: eager launches from now on, say why once
                self.failed = f"{type(e).__name__}: {str(e)[:200]}"
                self.cache.clear()
                import warnings
                warnings.warn(f"omni3d_amd: hipGraph capture of the inference pass failed ({self.failed}); running eager launches")
                return None
            self.cache[sig] = entry
            while len(self.cache) > max(CACHE, 1):
                self.cache.popitem(last=False)
        else:
            self._stage(entry, batched_inputs)
        self.cache.move_to_end(sig)
        if entry["graph"] is not None:
            entry["graph"].replay()
        else:
            with torch.no_grad(), context():
                entry["raw"] = self.model._inference_device(entry["batch"], entry["packed"])
        self.replays += 1
        sizes = [(b["image"].shape[-2], b["image"].shape[-1]) for b in batched_inputs]
        # (copies: the results handed to the caller must not alias the tensors the next replay overwrites -- a few hundred KB)
        return {k: v.clone() for k, v in entry["raw"].items()}, sizes

    def signature(self, batch):
        """(batch size, padded height, padded width) on the grid the EAGER pass pads to (`backbone.size_divisibility`, 64 for the
        reference's configurations): a replayed pass sees the tensor an eager pass would have built"""
        from ...solver.autoreplay import AutoReplay
        g = AutoReplay._model_bucket(self.model)
        H = max(b["image"].shape[-2] for b in batch)
        W = max(b["image"].shape[-1] for b in batch)
        return (len(batch), -(-H // g) * g, -(-W // g) * g)

    def _digest(self):
        """changes whenever a captured pass could be stale: a write torch knows about (`_version`), one it cannot see (PARAM_EPOCH:
        the flat optimizers and replayed training steps move values through raw pointers), or a parameter / buffer whose STORAGE was
        replaced (`model.to()`, `.double()`: the graph would keep reading the freed allocation) -- ADVICE r4"""
        from ..layers import PARAM_EPOCH
        # The tensor list is cached; `model.to()` / `.double()` / `.cuda()` REPLACE the buffer objects (and every parameter's storage), which
        # the cached list would not see.  No hook on `_apply` (a closure in the instance dict broke torch.save(model) and made
        # a deepcopy move the ORIGINAL's tensors) and no re-listing per call either (370 us of a 6.2 ms inference pass: 654 -> 629
        # images/s, measured round 6): any such conversion also moves the FIRST parameter's storage, so its (pointer, dtype, device) is the
        # fingerprint that decides whether the list is rebuilt.
        p0 = next(self.model.parameters(), None)
        fp = (p0.data_ptr(), p0.dtype, p0.device) if p0 is not None else None
        if self._tensors is None or fp != getattr(self, "_fp", None):
            self._tensors = list(self.model.parameters()) + list(self.model.buffers())
            self._fp = fp
        tensors = self._tensors
        return (PARAM_EPOCH[0], sum(t._version for t in tensors), len(tensors), hash(tuple(t.data_ptr() for t in tensors)))

    def _pack(self, batch):
        sizes = [(b["image"].shape[-2], b["image"].shape[-1]) for b in batch]
        return pack_targets(batch, sizes, getattr(self.model.roi_heads, "virtual_focal", 512.0), with_gt=False)

    def _capture(self, batch, sig, context):
        model, dev = self.model, self.model.device
        B, Hb, Wb = sig[:3]
        slots = torch.zeros((B, 3, Hb, Wb), dtype=batch[0]["image"].dtype, device=dev)
        sb = []
        for n, b in enumerate(batch):
            c = {k: v for k, v in b.items() if k not in ("image", "instances")}
            h, w = b["image"].shape[-2:]
            slots[n, :, :h, :w].copy_(b["image"], non_blocking=True)
            c["image"] = slots[n]
            sb.append(c)
        packed = self._pack(batch).to(dev)
        packed.slotted = True
        if self.graphs is False or dev.type != "cuda":
            self.captures += 1
            return {"graph": None, "raw": None, "slots": slots, "batch": sb, "packed": packed}
        from ....kernels import detmode
        detmode.prewarm(dev, ("M",))                # the graph family's arrival counters exist BEFORE the capture: a first use inside it would
                                                    # put a 256 KB zero-fill node (and pool memory) behind every deterministic launch
        if self._side is None:
            self._side = torch.cuda.Stream()        # (one for all captures: per-stream state elsewhere -- arrival counters -- is keyed by it)
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad(), context():
            model._inference_device(sb, packed)               # per-shape caches (anchors, index vectors) exist before the capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"), torch.no_grad(), context():
            raw = model._inference_device(sb, packed)
        self.captures += 1
        return {"graph": g, "raw": raw, "slots": slots, "batch": sb, "packed": packed}

    def _stage(self, entry, batch):
        dev = self.model.device
        for n, b in enumerate(batch):
            h, w = b["image"].shape[-2:]
            entry["slots"][n, :, :h, :w].copy_(b["image"], non_blocking=True)       # (the rest of the slot is masked by image_hw)
        new, sp = self._pack(batch), entry["packed"]
        for f in FIELDS:
            getattr(sp, f).copy_(getattr(new, f).to(dev, non_blocking=True), non_blocking=True)
