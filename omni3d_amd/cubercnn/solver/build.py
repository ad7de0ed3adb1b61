"""`build_optimizer(cfg, model)` / `freeze_bn` with the reference's parameter-group rules
(/root/reference/cubercnn/solver/build.py:6-76): SGD momentum (cfg.SOLVER.MOMENTUM / NESTEROV) or Adam / AdamW with or without
amsgrad (SOLVER.TYPE, eps 1e-2 as the reference passes), weight decay WEIGHT_DECAY except WEIGHT_DECAY_NORM for norm layers,
WEIGHT_DECAY_BIAS / BIAS_LR_FACTOR for biases, 0 for the `priors_*` parameters.

MI355X design: instead of ~230 per-tensor update kernels, every parameter is re-homed into ONE flat fp32
bucket (grouped by weight decay so a group is a contiguous range), gradients accumulate into a matching flat
bucket (`param.grad` are views, so the training script's per-parameter NaN scan at tools/train_net.py:226-233
still works), and `step()` is one fused kernel per group.  The flat gradient bucket is also what the
data-parallel all-reduce operates on (one RCCL call, no per-tensor launches), and a fused non-finite scan of
it can gate the step on the device."""
import os

import torch

from ...kernels import det

CL = torch.channels_last


def _param_groups(cfg, model):
    norm_types = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm, torch.nn.GroupNorm,
                  torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d, torch.nn.LayerNorm,
                  torch.nn.LocalResponseNorm)
    groups, memo = [], set()
    for module in model.modules():
        for key, value in module.named_parameters(recurse=False):
            if not value.requires_grad or value in memo or key == "_omni_ddp_anchor":     # the anchor belongs to DDP (solver/ddp.py)
                continue
            memo.add(value)
            lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
            if isinstance(module, norm_types) and cfg.SOLVER.WEIGHT_DECAY_NORM is not None:
                wd = cfg.SOLVER.WEIGHT_DECAY_NORM
            elif key == "bias":
                if cfg.SOLVER.BIAS_LR_FACTOR is not None:
                    lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
                if cfg.SOLVER.WEIGHT_DECAY_BIAS is not None:
                    wd = cfg.SOLVER.WEIGHT_DECAY_BIAS
            if key in ("priors_dims_per_cat", "priors_z_scales", "priors_z_stats"):
                wd = 0.0
            groups.append({"params": [value], "lr": lr, "weight_decay": wd})
    return groups


class _FlatOptimizer(torch.optim.Optimizer):
    """Flat parameter / gradient / state buckets, the data-parallel exchange over them and the re-binding of stray gradients:
    everything the fused optimizers share.  Subclasses name their per-element state buckets in STATE and implement
    `_step_segments()`."""
    STATE = ()

    def __init__(self, params, defaults, direct_accumulate=True):
        super().__init__(params, defaults)
        self._build_buckets()
        # backward kernels add weight / bias / BN gradients straight into the bucket views (functional._direct_grad).
        # Must be off under torch DistributedDataParallel, whose reducer needs autograd's accumulation hooks.
        self.set_direct_accumulate(direct_accumulate)
        self._steps = 0
        self.skip_flag = None   # optional device float: != 0 skips the update inside the kernel
        self._grad_scale = 1.0  # consumed by the next step(): 1/world when the all-reduce left SUMS in the bucket
        self._exchanged = False  # did this step's gradients go through all_reduce_begin / all_reduce_finish?
        self._replay_state = None  # set by AutoReplay.forward: {"pending": all-reduce handles, "zero_grads_seen": n}
        # step() averages the bucket itself when a process group with > 1 rank exists and nobody exchanged this step's
        # gradients.  build_optimizer switches it off when somebody else owns the exchange (a DistributedDataParallel wrapper
        # whose reducer already averaged the gradients: a second pass over the 191.6 MB bucket would only cost time)
        self.exchange_in_step = True

    def _build_buckets(self):
        plist = [(g, p) for g in self.param_groups for p in g["params"]]
        if not plist:
            raise ValueError("optimizer got an empty parameter list")
        dev = plist[0][1].device
        # contiguous range per (lr / base lr, weight decay) class; 16-byte aligned segments
        classes = {}
        for g, p in plist:
            classes.setdefault((float(g["lr"]), float(g["weight_decay"])), []).append((g, p))
        # Fused parameter groups (`tag_fused_groups`): parameters a module evaluates as ONE matrix -- cls_score + bbox_pred of the box
        # predictor, the five bbox_3D_* heads of the cube head (+ zero rows up to the kernels' multiple of 16) -- sit back to back in
        # that order, followed by their zero padding, so the fused matrix and its gradient are plain views of the buckets: no
        # torch.cat per forward, one weight-gradient launch that accumulates in place, no per-member gradient adds.
        layout = {}
        total = 0
        for key, items in classes.items():
            items = sorted(items, key=lambda gp: -self._grad_stage(gp[1]))    # stable: [last backward stage | ... | stage 1 | heads]
            ids = {id(p) for _, p in items}
            entries, done = [], set()
            for g, p in items:
                if id(p) in done:
                    continue
                tag = getattr(p, "_omni_fuse", None)
                members = tag["members"] if tag is not None else None
                if members is not None and all(id(m) in ids for m in members) and not any(id(m) in done for m in members) and \
                        len({self._grad_stage(m) for m in members}) == 1:
                    gof = {id(pp): gg for gg, pp in items}
                    for m in members:
                        entries.append((gof[id(m)], m, m.numel(), tag))
                        done.add(id(m))
                    entries.append((None, None, int(tag["pad"]), tag))                      # zero padding behind the last member
                    entries.append((None, None, (-sum(m.numel() for m in members) - int(tag["pad"])) % 4, None))   # re-align
                else:
                    entries.append((g, p, p.numel(), None))
                    entries.append((None, None, (-p.numel()) % 4, None))
                    done.add(id(p))
            layout[key] = (items, entries)
            total += sum(e[2] for e in entries)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_state = {name: torch.zeros(total, dtype=torch.float32, device=dev) for name in self.STATE}
        # (start, end, index of a representative group in self.param_groups).  An INDEX, not the dict: torch's
        # Optimizer.load_state_dict replaces the group dicts, and the LR scheduler then writes to the new ones
        self.segments = []
        self._slot = {}      # id(param) -> (offset, numel) inside the flat buckets
        off = 0
        # Inside a class, parameters whose gradients complete EARLY in backward (the heads: everything that is not the
        # backbone, tagged `_omni_early_grad` by build_optimizer) sit behind the late ones, so each class is
        # [late | early] and the data-parallel exchange can all-reduce the early ranges while the backbone is still
        # back-propagating (all_reduce_begin / all_reduce_finish).
        # Round 4: one exchange range per BACKWARD STAGE and class (`_omni_grad_stage`, tagged by build_optimizer from the backbone's
        # cut points: 0 = heads, 1 = FPN + the deepest levels, ... last = the first layer), ordered last stage first, so the
        # gradients of stage k are all-reduced behind that stage's weight-gradient graph while the stages below still back-propagate
        # (graphed.py); `early_ranges` (stage 0) / `late_ranges` (everything else) keep the two-phase interface.
        self.stage_ranges = {}
        for key, (items, entries) in layout.items():
            start = off
            fused_start = {}
            cur_stage, cur_start = None, off
            for g, p, n, tag in entries:
                if p is None:
                    if tag is not None:              # the padding closes a fused group: hand its members the fused views
                        o = fused_start.pop(id(tag))
                        tag["members"][0]._omni_fused_view = (self.flat_param[o:off + n], self.flat_grad[o:off + n],
                                                              tuple(m.data_ptr() for m in tag["members"]))
                    off += n
                    continue
                st = self._grad_stage(p)
                if st != cur_stage:
                    if cur_stage is not None and off > cur_start:
                        self.stage_ranges.setdefault(cur_stage, []).append((cur_start, off))
                    cur_stage, cur_start = st, off
                if tag is not None and id(tag) not in fused_start:
                    fused_start[id(tag)] = off
                pv = self._view_like(self.flat_param[off:off + n], p)
                pv.copy_(p.data)
                p.data = pv
                gv = self._view_like(self.flat_grad[off:off + n], p)
                if p.grad is not None:
                    gv.copy_(p.grad)
                p.grad = gv
                self._slot[id(p)] = (off, n)
                off += n
            if cur_stage is not None and off > cur_start:
                self.stage_ranges.setdefault(cur_stage, []).append((cur_start, off))
            self.segments.append((start, off, next(i for i, g in enumerate(self.param_groups) if g is items[0][0])))
        self.n_stages = (max(self.stage_ranges) + 1) if self.stage_ranges else 1
        self.early_ranges = list(self.stage_ranges.get(0, []))
        self.late_ranges = [r for k in sorted(self.stage_ranges) if k > 0 for r in self.stage_ranges[k]]

    @staticmethod
    def _grad_stage(p):
        st = getattr(p, "_omni_grad_stage", None)
        if st is not None:
            return int(st)
        return 0 if getattr(p, "_omni_early_grad", False) else 1

    def set_direct_accumulate(self, flag):
        self._direct = bool(flag)
        for g in self.param_groups:
            for p in g["params"]:
                p._omni_direct_grad = bool(flag)

    @staticmethod
    def _view_like(flat, p):
        if p.dim() == 4 and p.is_contiguous(memory_format=CL) and not p.is_contiguous():
            K, C, R, S = p.shape
            return flat.view(K, R, S, C).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    def zero_grad(self, set_to_none=False):
        """param.grad stay views of the flat bucket (set_to_none is ignored: the views ARE the storage the fused step and
        the all-reduce operate on)."""
        st = getattr(self, "_replay_state", None)
        if st is not None:        # model(data) replayed the whole step (cubercnn/solver/autoreplay.py): the bucket holds THIS
            st["zero_grads_seen"] += 1        # iteration's gradients; the loop's zero_grad() before backward() must not wipe them
            if st["zero_grads_seen"] == 1:
                return
            self._finish_replay_exchange()    # a second zero_grad(): the loop drops the iteration (tools/train_net.py:245-247)
        from ... import functional as HF
        HF.side_join()            # weight-gradient stream (normally already joined by the end-of-backward callback)
        from ... import lib as _lib
        from ...kernels import glue
        if self.flat_grad.is_cuda or (_lib._lib is not None and _lib._lib.emulated):
            glue.zero_(self.flat_grad)        # (a kernel node of this library: round 6, no ATen launch left in the step)
        else:
            self.flat_grad.zero_()
        self._exchanged = False
        # the bucket is empty again: a 1/world left behind by an exchange whose step() never came (the dropped iteration of
        # tools/train_net.py:245-247) must not scale the NEXT iteration's, separately averaged, gradients
        self._grad_scale = 1.0

    def _finish_replay_exchange(self):
        """pending all-reduce handles of a replayed step -> waited for, 1/world deferred into the update"""
        st, self._replay_state = getattr(self, "_replay_state", None), None
        if st is not None and st["pending"]:
            self.all_reduce_finish(st["pending"], defer_scale=True)
        return st is not None

    @torch.no_grad()
    def _rebind_grads(self):
        """`nn.Module.zero_grad()` (set_to_none=True by default), `p.grad = None` or `model.to()` detach a parameter's grad
        from the bucket; autograd then accumulates into a fresh tensor the fused step would never see.  Before every step:
        a stray gradient is copied into its bucket slot and re-bound; a missing one (None) means "no gradient this step",
        so its slot is cleared.  Costs one pointer compare per parameter."""
        fixed = 0
        # after a replayed model(data) the iteration's gradients exist ONLY in the bucket: a `model.zero_grad()` (set_to_none) between
        # the replay and step() detached the views but must not cost the slot its content; only optimizer.zero_grad()
        # called twice drops a replayed iteration
        replayed = getattr(self, "_replay_state", None) is not None
        bound = self.__dict__.setdefault("_bound_views", {})       # id(p) -> the bucket view this method (or __init__) bound as p.grad
        if self.__dict__.get("_bound_base") != self.flat_grad.data_ptr():
            bound.clear()
            self.__dict__["_bound_base"] = self.flat_grad.data_ptr()
        for g in self.param_groups:
            for p in g["params"]:
                cur = p.grad
                if cur is not None and cur is bound.get(id(p)):
                    continue                                        # (identity: the common case costs no tensor call)
                off, n = self._slot[id(p)]
                want = self.flat_grad[off:off + n]
                if cur is not None and cur.data_ptr() == want.data_ptr():
                    bound[id(p)] = cur
                    continue
                gv = self._view_like(want, p)
                if cur is None:
                    if not replayed:
                        gv.zero_()
                else:
                    gv.copy_(cur)
                p.grad = gv
                bound[id(p)] = gv
                fixed += 1
        if fixed:
            self.set_direct_accumulate(self._direct)
        return fixed

    @torch.no_grad()
    def all_reduce_grads(self, group=None):
        """Data-parallel exchange step, non-overlapped form: the early ranges then the late ranges (the same collective
        sequence as the overlapped form, so ranks may mix the two), then the average -- replaces DDP's per-bucket
        reducer for the native path (RCCL over xGMI on MI355X; gloo in the CPU tests)."""
        if getattr(self, "_replay_state", None) is not None:     # a replayed step started its exchange itself
            self._finish_replay_exchange()
            self._replay_state = {"pending": [], "zero_grads_seen": 1}
            return
        self.all_reduce_finish(self.all_reduce_begin("early", group) + self.all_reduce_begin("late", group), group)

    # elements per all-reduce call (default 32 MB): a stage's range is issued in pieces RCCL can pipeline.  OMNI_EXCHANGE_CHUNK_MB and
    # OMNI_EXCHANGE_MERGE_FROM are the two knobs the first multi-GPU run may want to sweep: xGMI rings are per-link
    # bound, so the best piece size is a property of the node, and the last stages' ranges are tiny (0.5 / 0.03 / 0.01 MB: pure latency) --
    # MERGE_FROM = k sends the ranges of stages k, k + 1, ... together behind the LAST stage instead of one call sequence per stage.
    # Both must be the same on every rank (they are part of the call sequence); the defaults are the round-4 behaviour.
    EXCHANGE_CHUNK = max(1, int(float(os.environ.get("OMNI_EXCHANGE_CHUNK_MB", "32")) * (1 << 18)))
    # Round 6 default "auto" (-2): the trailing stages whose ranges together stay below EXCHANGE_TAIL_MB (2 MB: DLA-34's stages 4-6 hold
    # 0.5 / 0.03 / 0.01 MB -- three latency-bound call sequences at the very end of backward) leave as ONE sequence behind the last
    # stage, adjacent ranges coalesced into one call; -1 = one sequence per stage (rounds 4-5), k >= 0 = merge from stage k.
    EXCHANGE_MERGE_FROM = int(os.environ.get("OMNI_EXCHANGE_MERGE_FROM", "-2"))
    EXCHANGE_TAIL_MB = float(os.environ.get("OMNI_EXCHANGE_TAIL_MB", "2"))
    exchange_timing = False         # bench.py, N > 1: device-side events around the waits of all_reduce_finish (exchange_report)

    def exchange_stages(self, k):
        """backward stages whose ranges go out behind stage k (see EXCHANGE_MERGE_FROM)"""
        m, last = self.merge_from(), self.n_stages - 1
        if m < 0 or m >= last or k < m:
            return [k]
        return list(range(m, self.n_stages)) if k == last else []

    def merge_from(self):
        """the stage from which the ranges leave together behind the last stage (-1: none); a function of the bucket layout and the
        environment knobs alone -- the same on every rank"""
        m = self.EXCHANGE_MERGE_FROM
        if m != -2:
            return m
        cached = self.__dict__.get("_merge_from_auto")
        if cached is None:
            limit, tail, k = self.EXCHANGE_TAIL_MB * (1 << 18), 0, self.n_stages
            while k > 1:        # (stage 0, the heads, is never part of the tail)
                size = sum(e - s for s, e in self.stage_ranges.get(k - 1, []))
                if tail + size > limit:
                    break
                tail += size
                k -= 1
            cached = self.__dict__["_merge_from_auto"] = k if k < self.n_stages - 1 else -1
        return cached

    def exchange_chunks(self, stages):
        """-> [(start, end)] of the given backward stages, stage by stage, class by class, cut into EXCHANGE_CHUNK pieces.  A function
        of the bucket layout (and the two environment knobs) alone: every rank issues the same sequence whether it replays a captured
        step or runs eager launches."""
        # The call boundaries must not depend on HOW a rank got here (stage by stage behind its weight-gradient graphs, or "early" +
        # "late" from an eager step): stages below the merge point are always sequences of their own; the stages from the merge point
        # on form ONE group whose ranges are coalesced where they touch in the bucket -- whenever all of them are asked for together,
        # which is the only way either form asks for them.
        m = self.merge_from()
        tail = [k for k in stages if m >= 0 and k >= m]
        groups = [[k] for k in stages if not (m >= 0 and k >= m)]
        if tail:
            groups.append(tail if tail == list(range(m, self.n_stages)) else None)
            if groups[-1] is None:
                groups = groups[:-1] + [[k] for k in tail]
        out = []
        for grp in groups:
            ranges = [r for k in grp for r in self.stage_ranges.get(k, [])]
            if len(grp) > 1:
                merged = []
                for s, e in sorted(ranges):
                    if merged and merged[-1][1] == s:
                        merged[-1] = (merged[-1][0], e)
                    else:
                        merged.append((s, e))
                ranges = merged
            for s, e in ranges:
                while s < e:
                    out.append((s, min(e, s + self.EXCHANGE_CHUNK)))
                    s += self.EXCHANGE_CHUNK
        return out

    def all_reduce_begin(self, which, group=None):
        """Starts the asynchronous all-reduce (sum) of a part of the flat gradient bucket and returns the pending work handles.
        which: "early" = backward stage 0 (every parameter outside the backbone: final once the heads have back-propagated), "late" =
        every other stage, an int k = stage k, ("from", k) = stages k, k + 1, ...  cubercnn/solver/graphed.py issues stage k behind
        that stage's weight-gradient graph, so RCCL moves each stage's bytes while the stages below still back-propagate."""
        import torch.distributed as dist
        if getattr(self, "_exchange_muted", False):        # warm-up passes of a staged-graph capture (graphed.py): rank-local
            return []
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return []
        self._exchanged = True
        if which == "early":
            stages = [0]
        elif which == "late":
            stages = list(range(1, self.n_stages))
        elif isinstance(which, tuple):
            stages = list(range(int(which[1]), self.n_stages))
        else:
            stages = self.exchange_stages(int(which))
        return [(dist.all_reduce(self.flat_grad[s:e], group=group, async_op=True), s, e) for s, e in self.exchange_chunks(stages)]

    def all_reduce_finish(self, pending, group=None, defer_scale=False):
        """Waits for the handles of all_reduce_begin (stream-ordered for RCCL) and averages.  defer_scale=True leaves the
        SUMS in the bucket and folds 1/world into the next step()'s kernel (saves one pass over the 191.6 MB bucket; the
        per-parameter `.grad` views then hold sums until the step -- the training loop only scans them for NaN/Inf)."""
        import torch.distributed as dist
        if not pending:
            return
        scale = 1.0 / dist.get_world_size(group)
        timed = self.exchange_timing and self.flat_grad.is_cuda
        if timed:       # how long the consuming stream really waits for the exchange = the part of it that did NOT overlap with backward
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for work, s, e in pending:
            work.wait()
            if not defer_scale:
                self.flat_grad[s:e].mul_(scale)
        if timed:
            ev[1].record()
            self._exchange_events = (getattr(self, "_exchange_events", []) + [(ev, len(pending), sum(e - s for _, s, e in pending))])[-64:]
        if defer_scale:
            self._grad_scale = scale

    def exchange_report(self, last=20):
        """-> what the timed steps' exchanges looked like from the consuming stream (None if nothing was timed)"""
        evs = getattr(self, "_exchange_events", [])[-last:]
        if not evs:
            return None
        torch.cuda.synchronize()
        return {"exposed_ms": sum(a.elapsed_time(b) for (a, b), _, _ in evs) / len(evs), "steps": len(evs),
                "all_reduce_calls_per_step": evs[-1][1], "bytes_per_step": 4 * evs[-1][2], "chunk_mb": self.EXCHANGE_CHUNK / (1 << 18),
                "merge_from_stage": self.merge_from(),
                "stage_range_mb": {k: round(4 * sum(e - s for s, e in v) / 1e6, 3) for k, v in sorted(getattr(self, "stage_ranges", {}).items())},
                "note": "exposed_ms: device time between the events around the waits of all_reduce_finish -- the share of the exchange that "
                        "backward did not hide"}

    @torch.no_grad()
    def check_nonfinite(self, flag):
        """flag (1,) device float: set to 1 if any gradient element is NaN/Inf (tools/train_net.py:222-233)."""
        det.nonfinite_any(self.flat_grad, flag)

    @torch.no_grad()
    def step(self, closure=None):
        from ... import functional as HF
        HF.side_join()
        self._rebind_grads()
        user_skip = self.skip_flag
        if self._finish_replay_exchange():
            auto = getattr(self, "_auto", None)
            if auto is not None and auto.bad is not None:
                # the fused update also skips when backward() saw a non-unit upstream gradient (autoreplay.py)
                self.skip_flag = auto.bad if user_skip is None else torch.add(user_skip, auto.bad)
                auto.arm_check()
        if not self._exchanged and self.exchange_in_step:
            # Data-parallel safety net.  The reference's loop (tools/train_net.py:449-454) relies on DistributedDataParallel's
            # autograd hooks for the gradient exchange; with direct accumulation the weight gradients never pass through
            # autograd, so those hooks would see nothing.  If the process group has more than one rank and nobody called
            # all_reduce_begin for this step, the bucket is averaged here (a no-op on values DDP already made identical).
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.all_reduce_grads()
        self._step_segments()
        from ..modeling.layers import PARAM_EPOCH
        PARAM_EPOCH[0] += 1                  # (inference-side caches of values derived from parameters: layers.BatchNorm2d)
        self.skip_flag = user_skip
        self._grad_scale = 1.0
        self._steps += 1

    def _ordered_params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _state_view(self, name, p):
        off, n = self._slot[id(p)]
        return self._view_like(self.flat_state[name][off:off + n], p)


class FlatSGD(_FlatOptimizer):
    """torch.optim.SGD semantics over flat buckets (see module docstring)."""
    STATE = ("momentum_buffer",)

    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, direct_accumulate=True):
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults, direct_accumulate)

    @property
    def flat_mom(self):
        return self.flat_state["momentum_buffer"]

    def _step_segments(self):
        first = self._steps == 0
        for start, end, gi in self.segments:
            g = self.param_groups[gi]          # looked up at step time: hyper-parameters are whatever the live groups say now
            det.sgd_step(self.flat_param[start:end], self.flat_grad[start:end], self.flat_mom[start:end], g["lr"],
                         g["momentum"], g["dampening"], g["weight_decay"], g["nesterov"], first_step=first, skip_flag=self.skip_flag,
                         grad_scale=self._grad_scale)

    # ---- checkpoint format: torch.optim.SGD's (per-parameter `momentum_buffer`), so optimizer states written by the
    # reference's DetectionCheckpointer (tools/train_net.py:128) load here and vice versa ----------------------------------
    def state_dict(self):
        sd = super().state_dict()          # param_groups with params packed as 0..N-1 in group order, like torch.optim.SGD
        state = {}
        if self._steps > 0:
            for idx, p in enumerate(self._ordered_params()):
                off, n = self._slot[id(p)]
                state[idx] = {"momentum_buffer": self._view_like(self.flat_mom[off:off + n], p).detach().clone()
                              .contiguous(memory_format=torch.contiguous_format)}
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        legacy = sd.pop("flat_momentum", None)           # round-1 layout-dependent format, still readable
        steps = sd.pop("steps", None)
        state = sd.get("state", {})
        params = self._ordered_params()
        sd["state"] = {}
        super().load_state_dict(sd)                      # learning rates, momentum, weight decay of the groups
        self.flat_mom.zero_()
        loaded = 0
        for idx, st in state.items():
            buf = st.get("momentum_buffer") if isinstance(st, dict) else None
            if buf is None:
                continue
            p = params[int(idx)]
            off, n = self._slot[id(p)]
            if buf.numel() != n:
                raise ValueError(f"momentum_buffer {idx}: {buf.numel()} elements, the parameter has {n}")
            self._view_like(self.flat_mom[off:off + n], p).copy_(buf.reshape(p.shape).to(self.flat_mom.device))
            loaded += 1
        if legacy is not None and loaded == 0:
            self.flat_mom.copy_(legacy)
            loaded = len(params)
        if loaded not in (0, len(params)) :
            raise ValueError(f"optimizer state holds momentum for {loaded} of {len(params)} parameters")
        if loaded == 0 and any(g["momentum"] != 0 for g in self.param_groups) and (steps or 0) > 0:
            raise ValueError("optimizer state has no momentum buffers although momentum != 0 and steps > 0")
        self._steps = int(steps) if steps is not None else (1 if loaded else 0)


class FlatAdam(_FlatOptimizer):
    """torch.optim.Adam / AdamW (amsgrad optional) semantics over the flat buckets: one fused kernel per (lr, weight decay)
    range reads p, g, exp_avg, exp_avg_sq (and max_exp_avg_sq) once.  The step count is a device scalar advanced by a one-lane
    kernel that honours the divergence guard's skip flag, so a skipped iteration leaves the bias corrections where they were
    (the reference does not call step() on such an iteration, tools/train_net.py:245-253)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, decoupled=False, direct_accumulate=True):
        self.STATE = ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ())
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=bool(amsgrad))
        self.decoupled = bool(decoupled)
        super().__init__(params, defaults, direct_accumulate)
        self.dev_step = torch.zeros(1, dtype=torch.float32, device=self.flat_param.device)

    def _step_segments(self):
        det.adam_tick(self.dev_step, self.skip_flag)
        st = self.flat_state
        for start, end, gi in self.segments:
            g = self.param_groups[gi]
            vmax = st["max_exp_avg_sq"][start:end] if g["amsgrad"] else None
            det.adam_step(self.flat_param[start:end], self.flat_grad[start:end], st["exp_avg"][start:end], st["exp_avg_sq"][start:end],
                          vmax, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.decoupled, self.dev_step,
                          skip_flag=self.skip_flag, grad_scale=self._grad_scale)

    # ---- checkpoint format: torch.optim.Adam's / AdamW's per-parameter {step, exp_avg, exp_avg_sq[, max_exp_avg_sq]} --------
    def state_dict(self):
        sd = super().state_dict()
        state = {}
        steps = float(self.dev_step.item())
        if steps > 0:
            for idx, p in enumerate(self._ordered_params()):
                entry = {"step": torch.tensor(steps)}
                for name in self.STATE:
                    entry[name] = self._state_view(name, p).detach().clone().contiguous(memory_format=torch.contiguous_format)
                state[idx] = entry
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        state = sd.get("state", {})
        params = self._ordered_params()
        sd["state"] = {}
        super().load_state_dict(sd)
        for buf in self.flat_state.values():
            buf.zero_()
        steps = set()
        for idx, st in state.items():
            p = params[int(idx)]
            steps.add(float(st["step"]))
            for name in self.STATE:
                if name not in st:
                    raise ValueError(f"optimizer state of parameter {idx} has no '{name}' (amsgrad mismatch?)")
                if st[name].numel() != p.numel():
                    raise ValueError(f"{name} {idx}: {st[name].numel()} elements, the parameter has {p.numel()}")
                self._state_view(name, p).copy_(st[name].reshape(p.shape).to(self.flat_param.device))
        if len(state) not in (0, len(params)):
            raise ValueError(f"optimizer state covers {len(state)} of {len(params)} parameters")
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}); the fused step keeps one count")
        self.dev_step.fill_(steps.pop() if steps else 0.0)
        self._steps = int(self.dev_step.item())


def tag_fused_groups(model):
    """modules that evaluate several parameters as one matrix name them (`fused_param_groups() -> [(members in fused order, zero
    elements of padding behind them)]`); the tags are read by _FlatOptimizer._build_buckets"""
    for m in model.modules():
        groups = getattr(m, "fused_param_groups", None)
        if groups is None:
            continue
        for members, pad in groups():
            members = list(members)
            for p in members:
                p._omni_fuse = None
            members[0]._omni_fuse = {"members": members, "pad": int(pad)}
            for p in members[1:]:
                p._omni_fuse = members[0]._omni_fuse


def fused_view(members, training):
    """-> (matrix storage, gradient storage or None) = flat views over `members` + their zero padding when the optimizer laid them out
    back to back (tag_fused_groups) and nothing has moved them since; None -> the caller concatenates.  The gradient storage is only
    handed out when the backward kernels may accumulate into the bucket directly (functional._direct_grad)."""
    fv = getattr(members[0], "_omni_fused_view", None)
    if fv is None:
        return None
    pflat, gflat, ptrs = fv
    if any(m.data_ptr() != q for m, q in zip(members, ptrs)):
        return None                                         # (model.to(), a re-assigned .data: the layout is gone)
    if not training:
        return pflat, None
    if not all(getattr(m, "_omni_direct_grad", False) and m.requires_grad and m.grad is not None for m in members):
        return None
    if members[0].grad.data_ptr() != gflat.data_ptr():
        return None
    return pflat, gflat


def grad_stage_map(model):
    """parameter name -> backward stage in which its gradient completes (cubercnn/solver/graphed.py GraphedPipelined): 0 = everything
    outside the backbone, 1 = FPN + the bottom-up levels above the last cut, 2, 3, ... = the pieces between the cuts going down, as
    the bottom-up declares them (`backward_stages()`: {name prefix: stage}); a backbone without cut points is stage 1 as a whole."""
    bu = getattr(getattr(model, "backbone", None), "bottom_up", None)
    table = bu.backward_stages() if (bu is not None and hasattr(bu, "backward_stages")) else {}
    # the cut at the pooled ROI features (graphed.py POOL_CUT) makes the FC heads stage 0 and everything whose gradient completes
    # with ROIAlign's / the RPN's backward stage 1; the backbone's stages move down by one
    shift = 1 if pool_cut_active(model) else 0

    def stage_of(name):
        if not name.startswith("backbone."):
            return shift if (shift and not name.startswith("roi_heads.")) else 0
        if name.startswith("backbone.bottom_up."):
            rest = name[len("backbone.bottom_up."):]
            for prefix, st in table.items():
                if rest == prefix or rest.startswith(prefix + "."):
                    return st + shift
        return 1 + shift
    return stage_of


def pool_cut_active(model):
    from .graphed import POOL_CUT
    heads = getattr(model, "roi_heads", None)
    return bool(POOL_CUT and heads is not None and hasattr(type(heads), "pool_cut") and hasattr(getattr(model, "proposal_generator", None), "forward"))


def build_optimizer(cfg, model):
    from ... import respect_cpu_quota
    respect_cpu_quota()       # (the loop's host side must not wake more threads than the container may run: omni3d_amd/__init__.py)
    params = _param_groups(cfg, model)
    tag_fused_groups(model.module if hasattr(model, "module") else model)
    # gradients of everything outside the backbone are complete before the backbone starts back-propagating
    inner = model.module if hasattr(model, "module") else model
    stage_of = grad_stage_map(inner)
    for name, p in inner.named_parameters():
        p._omni_early_grad = not name.startswith("backbone.")
        p._omni_grad_stage = stage_of(name)
    # tools/train_net.py:449-454 wraps the model in DistributedDataParallel BEFORE do_train builds the optimizer.  A wrapper that
    # reduces the gradients itself needs autograd's accumulation hooks (no direct accumulation) and makes the step's own exchange
    # redundant; a wrapper this package's build_model prepared (`_omni_owns_exchange`, cubercnn/solver/ddp.py) leaves the exchange
    # to the optimizer's overlapped two-range all-reduce and keeps direct accumulation.
    under_ddp = isinstance(model, torch.nn.parallel.DistributedDataParallel)
    own = bool(getattr(inner, "_omni_owns_exchange", False))
    direct = (not under_ddp) or own
    if cfg.SOLVER.TYPE == "sgd":
        opt = FlatSGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, nesterov=cfg.SOLVER.NESTEROV,
                      weight_decay=cfg.SOLVER.WEIGHT_DECAY, direct_accumulate=direct)
    elif cfg.SOLVER.TYPE in ("adam", "adam+amsgrad", "adamw", "adamw+amsgrad"):     # build.py:58-65
        adamw = cfg.SOLVER.TYPE.startswith("adamw")
        # torch defaults behind the reference's calls: betas (0.9, 0.999); the groups carry their own weight decay
        opt = FlatAdam(params, cfg.SOLVER.BASE_LR, eps=1e-02, weight_decay=1e-2 if adamw else 0.0, amsgrad=cfg.SOLVER.TYPE.endswith("+amsgrad"),
                       decoupled=adamw, direct_accumulate=direct)
    else:
        raise ValueError("{} is not supported as an optimizer.".format(cfg.SOLVER.TYPE))
    bu = getattr(getattr(inner, "backbone", None), "bottom_up", None)
    opt.stage_cut_signature = (tuple(getattr(bu, "stage_cut_at", ())) if bu is not None else ()) + (("pool",) if pool_cut_active(inner) else ())
    opt.exchange_in_step = direct          # DDP's reducer already averaged: never all-reduce the bucket a second time
    if direct:
        from .autoreplay import attach
        attach(model, opt)                 # model(data) switches to staged-graph replay once the batch signature repeats
    if under_ddp and own:
        from .ddp import broadcast_replica
        broadcast_replica(inner, opt)      # the initial parameter / buffer broadcast DDP's constructor skipped for ignored names
    return opt


def freeze_bn(network):
    for _, module in network.named_modules():
        if isinstance(module, torch.nn.BatchNorm2d):
            module.eval()
            module.track_running_stats = False
