"""torch.autograd.Function wrappers around the C-ABI kernels.

These are the only place where torch's autograd graph meets the HIP kernels: each Function calls
the forward launcher, keeps what its hand-written backward kernel needs, and returns gradients
computed by the backward kernels.  4-D activations / weights are logical NCHW / KCRS tensors in
channels_last memory (physically NHWC / KRSC).
"""
import torch
from torch.autograd import Function
from os import environ as _environ

_os_environ_get = _environ.get

from .kernels import glue
from .kernels import bnpool, conv, det, wino

CL = torch.channels_last


def _cl(t):
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def _direct_grad(t):
    """The parameter's view of the flat gradient bucket when the optimizer asked for direct accumulation
    (FlatSGD): backward kernels then ADD into it and return no gradient, which saves one elementwise add kernel
    per parameter per step (~220 launches).  None otherwise (plain autograd accumulation)."""
    if t is not None and getattr(t, "_omni_direct_grad", False) and t.grad is not None and t.requires_grad:
        return t.grad
    return None


def _parts_out(parts, like):
    """BatchNorm partial statistics as a second (non-differentiable) output; a 0-row tensor stands for "not produced" """
    return parts if parts is not None else like.new_zeros((0, 1))


# ---- gradient fan-in ---------------------------------------------------------------------------------------------------------
# An activation with several consumers (the input of a DLA Tree: max-pool + first block; a block output that is the next block's
# input, its residual and a Root child; an FPN top-down map; an FPN output read by the RPN head and by ROIAlign) has a gradient that
# is the SUM of its consumers' gradients, and the autograd engine forms that sum with one elementwise add kernel per extra
# consumer: ~80 launches and ~0.5 ms of a 12.6 ms DLA-34 step (profiles/r03_trace_table_final.txt).  `fanout(t)` hangs a slot on
# such a tensor; every backward function of this file that consumes it registers in its forward, and in backward
#   * reads what the consumers that ran before it left in the slot (`carry`) INSIDE its own kernel's epilogue (the *_carry entry
#     points of include/omni3d_hip.h, omni_conv2d_dgrad(accumulate = 1)) -- no separate add kernel, two reads and one write less;
#   * returns None to the engine unless it is the last registered consumer, which returns the complete sum.
# Consumers without a slot-aware kernel (ATen ops, anything outside this file) are unaffected: their gradient reaches the producer
# through the engine's own accumulation, on top of the one defined gradient the slot hands over.  A registered consumer whose
# backward never runs (its output does not reach the loss -- the projection a nested DLA Tree computes and drops, dla.py:208)
# would leave the sum withheld: a pre-hook on the producer's autograd node hands over whatever a slot still holds when the producer
# is about to run.  A leaf's consumers (the detached copies of solver/graphed.py's stage cuts) may run in different backward
# calls: the slot keeps the partial sum in between, and the cut adds `fanout_leftover` when it reads the leaf's gradient.
_FANOUT = _os_environ_get("OMNI_FANOUT", "1") != "0"          # A/B knob


class _GradSlot:
    __slots__ = ("buf", "remaining")

    def __init__(self):
        self.buf, self.remaining = None, 0


def fanout(t):
    """marks a tensor that several slot-aware consumers read (idempotent; a no-op outside training) -> t"""
    if _FANOUT and t.requires_grad and torch.is_grad_enabled() and getattr(t, "_omni_slot", None) is None:
        slot = t._omni_slot = _GradSlot()
        if t.grad_fn is not None:
            nr = t.output_nr

            def hand_over(grads):       # the producer is about to run: every consumer that will ever run has run
                left = fanout_leftover(slot)
                if left is None:
                    return None
                if grads[nr] is None:       # (autograd does not let a hook turn an undefined gradient into a defined one)
                    raise RuntimeError(f"omni3d_amd.functional.fanout: a consumer of a {tuple(left.shape)} activation registered with its "
                                       "gradient fan-in slot in forward but never ran in backward (its output does not reach the loss), and "
                                       "no other gradient arrived that the withheld sum could be added to.  Feed that consumer a detached "
                                       "input, or set OMNI_FANOUT=0.")
                grads = list(grads)
                grads[nr] = grads[nr] + left
                return tuple(grads)
            t.grad_fn.register_prehook(hand_over)
    return t


def fanout_leftover(t_or_slot):
    """-> the partial gradient sum a slot still withholds (a registered consumer never ran), or None; clears the slot"""
    slot = t_or_slot if isinstance(t_or_slot, _GradSlot) else getattr(t_or_slot, "_omni_slot", None)
    if slot is None or slot.buf is None:
        return None
    left, slot.buf, slot.remaining = slot.buf, None, 0
    return left


def _slot_enter(t, needs_grad):
    """forward of a consumer: register with the slot of input `t` (None: `t` has no slot / needs no gradient)"""
    slot = getattr(t, "_omni_slot", None) if (needs_grad and t is not None) else None
    if slot is not None:
        slot.remaining += 1
    return slot


def _slot_deliver(slot, compute):
    """backward of a consumer: compute(carry) -> this consumer's gradient PLUS carry (carry None: nothing to add).
    -> what the backward function returns for that input."""
    if slot is None:
        return compute(None)
    out = compute(slot.buf)
    slot.remaining -= 1
    if slot.remaining <= 0:
        slot.buf = None
        return out
    slot.buf = out
    return None


def _add_carry(out, carry):
    """consumers without a fan-in epilogue"""
    return out if carry is None else out + carry


def _carry_pitch(c):
    """-> pixel pitch (floats) of a logical (N,C,H,W) tensor that is NHWC in memory with channel stride 1 -- a whole channels_last
    tensor or a channel slice of a wider one --, or None when the kernels cannot read it as a carry"""
    N, C, H, W = c.shape
    sn, sc, sh, sw = c.stride()
    if c.dtype != torch.float32 or sc != 1 or sw < C or (sw & 3) or (c.data_ptr() & 15) or (c.storage_offset() & 3):
        return None
    if (H > 1 and sh != W * sw) or (N > 1 and sn != H * W * sw):
        return None
    return sw


# ---- weight gradients off the critical path ---------------------------------------------------------------------------------
# In the backward pass of a convolution / linear layer the data gradient is on the critical path (the next layer down waits for
# it) while the weight gradient is needed only by the optimizer, and most layers of this network launch too few workgroups to
# fill 256 CUs by themselves.  The backward functions therefore hand their weight-gradient launch (a closure that accumulates
# into the FlatSGD bucket in place and returns nothing) to `_side_run`, which does one of three things:
#   mode "inline"  : runs it on the spot (CPU / emulator, OMNI_WGRAD_STREAM=0, or no in-place gradient bucket);
#   mode "stream"  : eager steps -- queues it and launches every `batch` closures on a second HIP stream behind one
#                    cross-stream dependency; the main stream waits for the side stream once, when the backward pass ends
#                    (autograd engine callback), before anything reads the gradient bucket;
#   mode "collect" : hipGraph capture (cubercnn/solver/graphed.py GraphedPipelined) -- only queues; the capturer takes the
#                    closures of a backward stage and records them as a second graph that replays on the side stream next
#                    to the NEXT stage's data-gradient graph.  (A single captured graph with parallel branches is no use:
#                    ROCm 7.2 replays such a graph node by node from the host, 12 ms per step instead of 0.7 ms.)
# Inputs of queued closures are kept referenced until they have run and been joined, so the caching allocator cannot hand
# their memory to a main-stream kernel early.
_side = {"mode": "stream" if _os_environ_get("OMNI_WGRAD_STREAM", "1") != "0" else "inline",
         "batch": int(_os_environ_get("OMNI_WGRAD_BATCH", "8")), "streams": {}, "pending": [], "queue": [], "main": None}


def side_mode(mode=None):
    """-> current mode; sets it when given ("inline" | "stream" | "collect")"""
    if mode is not None:
        assert mode in ("inline", "stream", "collect")
        side_join()
        _side["mode"] = mode
    return _side["mode"]


def side_take():
    """collect mode: -> (queued closures, the tensors they read) and forget them"""
    q, keep = list(_side["queue"]), list(_side["pending"])
    _side["queue"].clear()
    _side["pending"].clear()
    return q, keep


def _side_flush():
    """launch the queued weight-gradient closures on the side stream behind ONE wait on the main stream"""
    if not _side["queue"]:
        return
    main, st = _side["main"]
    st.wait_stream(main)
    with torch.cuda.stream(st), wino.batched_wgrads():      # (the queued Winograd-domain GEMMs leave in one launch)
        for fn in _side["queue"]:
            fn()
    _side["queue"].clear()


def _side_run(fn, keepalive):
    dev = keepalive[0].device
    mode = _side["mode"]
    if mode == "inline" or (dev.type != "cuda" and mode != "collect"):
        return fn()
    if _DEBUG_DROP_SIDE:           # timing experiment only (tools): the critical path alone, parameter gradients never computed
        return None
    _side["queue"].append(fn)
    _side["pending"].extend(keepalive)
    if mode == "collect":
        return None
    if _side["main"] is None:
        st = _side["streams"].get(dev.index)
        if st is None:
            st = _side["streams"][dev.index] = torch.cuda.Stream(device=dev)
        _side["main"] = (torch.cuda.current_stream(dev), st)
        torch.autograd.Variable._execution_engine.queue_callback(side_join)    # runs when this backward pass ends
    if len(_side["queue"]) >= _side["batch"]:
        _side_flush()
    return None


# ---- a second branch inside the forward pass (round 4) ------------------------------------------------------------------------------
# The RPN's anchor labelling + sampling + loss forward (0.14 ms of latency-bound launches) and its proposal selection (top-k, decode,
# NMS, top-k: 0.32 ms of them) both start from the head outputs and do not read each other.  While a step is being captured
# (solver/graphed.py installs a stream) the first runs on that stream -- a parallel branch of the SAME hipGraph -- and the capturing
# stream waits for it where the proposals are done.  Without an installed stream (eager steps, tests) nothing changes.
_branch = {"stream": None, "open": False}


def set_branch_stream(stream):
    """-> the previous one; None switches the fork off"""
    prev, _branch["stream"], _branch["open"] = _branch["stream"], stream, False
    return prev


class forked:
    """with forked(): launches inside go to the branch stream (which first waits for everything the current stream has been given);
    join_branch() makes the current stream wait for the branch.  Custom Functions enter it INSIDE forward(), so that autograd keeps
    seeing the ambient stream and runs their backward there."""

    def __enter__(self):
        self.ctx = None
        st = _branch["stream"]
        if st is not None:
            if not _branch["open"]:
                st.wait_stream(torch.cuda.current_stream())
                _branch["open"] = True
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
        return False


def join_branch():
    if _branch["open"]:
        torch.cuda.current_stream().wait_stream(_branch["stream"])
        _branch["open"] = False


_DEBUG_DROP_SIDE = _environ.get("OMNI_DEBUG_DROP_SIDE", "0") == "1"
BIAS_GRAD_SIDE = _environ.get("OMNI_BIAS_GRAD_SIDE", "1") != "0"


def _bias_grad(dy2d, gb):
    """column sums of dy (P, C) -> the bias gradient.  With a view of the gradient bucket to add into, the two launches (partial
    rows + fixed-order finalize) go to the weight-gradient stream like the filter gradient: nothing on the critical path reads them
    (round 4: 44 launches, 0.27 ms per step, sat between the data-gradient kernels of the FPN / RPN / head layers)."""
    if gb is None or not BIAS_GRAD_SIDE:
        return bnpool.bias_grad(dy2d, accum_into=gb)
    return _side_run(lambda: bnpool.bias_grad(dy2d, accum_into=gb), (dy2d,))


def side_join():
    """stream mode: flush, then the main stream waits for the weight-gradient stream (idempotent; FlatSGD also calls it before
    it touches the gradient bucket).  collect mode: nothing to do, the capturer owns the queue."""
    if _side["mode"] == "collect":
        return
    if _side["main"] is not None:
        _side_flush()
        main, st = _side["main"]
        main.wait_stream(st)
        _side["main"] = None
    for fn in _side["queue"]:      # (queued without a stream: cannot happen in stream mode, kept for safety)
        fn()
    _side["queue"].clear()
    _side["pending"].clear()


def _relu_already_masked(dy):
    """The fused RPN head's data gradient applies the ReLU mask of the convolution below it on the way out and says so with a tag on
    the gradient tensor (`_omni_relu_masked` = the tensor's version counter at that moment).  The tag is only good while the tensor
    is what the head wrote: if the convolution's output ever gets a second consumer, the autograd engine may accumulate that consumer's
    (unmasked) gradient IN PLACE into the tagged tensor -- the version counter then differs, the sum can no longer be masked
    correctly, and this raises instead of returning wrong gradients."""
    tag = getattr(dy, "_omni_relu_masked", None)
    if tag is None or tag is False:
        return False
    if tag is not True and dy._version != tag:
        raise RuntimeError("omni3d_amd: the gradient of a convolution whose ReLU mask was applied by its consumer (fused RPN head) was "
                           "modified in place afterwards -- a second consumer of that convolution's output; set OMNI_RPN_HEAD16=0")
    return True



class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, relu, want_stats=False):
        ctx.set_materialize_grads(False)      # no zero tensors for the non-differentiable side outputs
        ctx.direct = (_direct_grad(w), _direct_grad(bias))
        ctx.slot = _slot_enter(x, ctx.needs_input_grad[0])
        x, w = _cl(x), _cl(w)
        # full-resolution few-channel stem layers: direct convolution with the input halo staged once in LDS
        ctx.stem = bias is None and not relu and conv.stem_eligible(x.shape, w.shape, stride, pad)
        parts = None
        if want_stats and bias is None and not relu:     # conv -> BatchNorm: statistics from the epilogue (saves a pass over y)
            y, parts = conv.stem_conv_fwd_stats(x, w) if ctx.stem else conv.conv2d_fwd_stats(x, w, stride, pad)
        else:
            y = conv.stem_conv_fwd(x, w) if ctx.stem else conv.conv2d_fwd(x, w, bias, stride, pad, relu)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (stride, pad, relu, bias is not None)
        parts = _parts_out(parts, y)
        ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, dy, _parts_grad=None):
        x, w, y = ctx.saved_tensors
        stride, pad, relu, has_bias = ctx.cfg
        relu = relu and not _relu_already_masked(dy)       # (the consumer's data-gradient kernel applied the mask)
        dy = _cl(dy)
        if relu:   # elementwise on the NHWC views (same dense layout for dy and y)
            dy = bnpool.relu_bwd(dy.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        gw, gb = ctx.direct
        if gw is not None and not gw.is_contiguous(memory_format=CL):
            gw = None
        dx = None
        if ctx.needs_input_grad[0]:
            def dgrad(carry):
                if conv.stem_dgrad_eligible(x.shape, w.shape, stride, pad):
                    # full-resolution stem layers: input halo + whole filter in LDS (csrc/stem_conv.hip); the stride-1 form is the
                    # same convolution of dy with the 180-degree rotated, channel-transposed filter, formed inside the kernel
                    return _add_carry(conv.stem_conv_dgrad(dy, w, (x.shape[2], x.shape[3]), stride), carry)
                if ctx.stem and w.shape[1] == 16:
                    return _add_carry(conv.stem_conv_fwd(dy, _cl(w.flip(2, 3).transpose(0, 1))), carry)
                if carry is not None and _carry_pitch(carry) is not None:
                    return conv.conv2d_dgrad(dy, w, (x.shape[2], x.shape[3]), stride, pad, accum_into=carry)
                return _add_carry(conv.conv2d_dgrad(dy, w, (x.shape[2], x.shape[3]), stride, pad), carry)
            dx = _slot_deliver(ctx.slot, dgrad)
        dw = None
        if ctx.needs_input_grad[1]:
            # (measured: the stem weight-gradient kernel wins for the 16-channel layer, 0.13 vs 0.21 ms, not for the
            # 4-channel 7x7 layer, 0.26 vs 0.21 ms, which keeps the split-K implicit GEMM)
            def wgrad():
                return (conv.stem_conv_wgrad(x, dy, w.shape[2], accum_into=gw, stride=stride) if conv.stem_wgrad_eligible(x.shape, w.shape, stride, pad)
                        else conv.conv2d_wgrad(x, dy, (w.shape[2], w.shape[3]), stride, pad, accum_into=gw))
            dw = _side_run(wgrad, (x, dy)) if gw is not None else wgrad()
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = _bias_grad(dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1]), gb)
        return dx, dw, db, None, None, None, None


_STEM_FIRST_WGRAD_SIDE = _os_environ_get("OMNI_STEM_FIRST_WGRAD_SIDE", "0") == "1"      # A/B knob


class _StemFirst(Function):
    """The first layer (dla.py:241-245: 7x7, 3 -> 16, stride 1) on the 4-channel padded image, with the 3-channel filter read and its
    gradient written in the model's own layout (csrc/stem_conv.hip, round 6).  The image needs no gradient."""

    @staticmethod
    def forward(ctx, x, w, want_stats):
        ctx.set_materialize_grads(False)
        ctx.direct = _direct_grad(w)
        x = _cl(x)
        y, parts = conv.stem_first_fwd(x, w, want_stats)
        ctx.save_for_backward(x)
        parts = _parts_out(parts, y)
        ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, dy, _parts_grad=None):
        (x,) = ctx.saved_tensors
        dy = _cl(dy)
        gw = ctx.direct
        if gw is not None and not gw.is_contiguous(memory_format=CL):
            gw = None
        dw = None
        if ctx.needs_input_grad[1]:
            def wgrad():
                return conv.stem_first_wgrad(x, dy, accum_into=gw)
            # This is the LAST weight gradient of a step (its dy is the last tensor backward produces).  Measured (round 6,
            # profiles/r06_ab_tail_balance.log): the weight-gradient stream reaches the end of backward ~0.15 ms behind the main
            # stream, so queued there this launch started 150 us after its input was ready; on the main stream it starts at once and
            # the two streams end together.
            dw = _side_run(wgrad, (x, dy)) if (gw is not None and _STEM_FIRST_WGRAD_SIDE) else wgrad()
        return None, dw, None


def stem_first_conv(x, w, want_stats=False):
    """conv2d(x[:, :3], w, padding=3) for the (N,4,H,W) padded image and the (16,3,7,7) filter; the caller checked
    `conv.stem_first_eligible(x.shape, w)`"""
    y, parts = _StemFirst.apply(x, w, want_stats)
    if parts.shape[0] > 0:
        y._omni_bn_partials = parts
    return y


class _WinoConv3x3(Function):
    """3x3 / stride 1 / pad 1 convolution through Winograd F(2x2,3x3) (csrc/winograd.hip): forward, data gradient (the same
    algorithm on dy with the rotated filter) and weight gradient (in the Winograd domain, reusing the forward's transformed
    input) each run 16 batched fp32-MFMA GEMMs with 2.25x fewer flops than the direct implicit GEMM."""

    @staticmethod
    def forward(ctx, x, w, bias, relu, want_stats=False):
        ctx.set_materialize_grads(False)      # no zero tensors for the non-differentiable side outputs
        ctx.direct = (_direct_grad(w), _direct_grad(bias))
        ctx.bn_below = getattr(x, "_omni_bn_below", None)      # x is the output of a BatchNorm(+ReLU) without residual
        ctx.slot = _slot_enter(x, ctx.needs_input_grad[0])
        w_given = w
        x, w = _cl(x), _cl(w)
        # one launch yields the forward transform U and (when the data gradient will also go through Winograd) U' of
        # the rotated filter; a weight shared by several calls of one step (the RPN conv over the FPN levels) is
        # transformed once (wino_weight_scope)
        need_flip = x.requires_grad and wino.dgrad_eligible(x.shape)
        tile = wino.tile_size(x.shape)
        # The cache only lives inside one model forward (RCNN3D.forward opens and closes it): nothing can change a weight
        # in between, and no stale entry can outlive the tensor it was computed from.
        key = (w.data_ptr(), tuple(w.shape), tile)
        cache = _wino_scope["cache"]
        U, Uf = cache.get(key, (None, None)) if cache is not None else (None, None)
        if U is None or (need_flip and Uf is None):
            U, Uf = wino.transform_weights(w, True, need_flip or Uf is not None, tile)
            if cache is not None:
                cache[key] = (U, Uf)
                if w is w_given:        # (a filter that had to be re-laid-out first is a temporary: nothing to remember)
                    _wino_scope["record"].append((w, tile, bool(need_flip or Uf is not None)))      # -> next pass: one batched launch
        parts = None
        if want_stats and bias is None and not relu:
            y, V, parts = wino.conv3x3_fwd(x, w, None, False, U=U, tile=tile, want_stats=True)
        else:
            y, V = wino.conv3x3_fwd(x, w, bias, relu, U=U, tile=tile)
        ctx.save_for_backward(V, w, y if relu else None, Uf if need_flip else None)
        ctx.cfg = (relu, bias is not None)
        parts = _parts_out(parts, y)
        ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, dy, _parts_grad=None):
        V, w, y, Uf = ctx.saved_tensors
        relu, has_bias = ctx.cfg
        relu = relu and not _relu_already_masked(dy)       # (the consumer's data-gradient kernel applied the mask)
        dy = _cl(dy)
        if relu:
            dy = bnpool.relu_bwd(dy.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        gw, gb = ctx.direct
        if gw is not None and not gw.is_contiguous(memory_format=CL):
            gw = None
        dx = dw = None
        slot, box = ctx.slot, []

        def fused(carry):       # carry: read by the output transform of the data gradient (omni_wino_out_carry)
            ok = carry is not None and _carry_pitch(carry) is not None
            dx_, dw_ = wino.conv3x3_backward(V, dy, w, Uf, accum_into=gw, side_run=_side_run, bn_below=ctx.bn_below if carry is None else None,
                                             carry=carry if ok else None)   # both transforms of dy in one pass
            box.append(dw_)
            return dx_ if (ok or carry is None) else dx_ + carry

        def dgrad_only(carry):
            if wino.dgrad_eligible(dy.shape):
                ok = carry is not None and _carry_pitch(carry) is not None
                dx_ = wino.conv3x3_dgrad(dy, w, U_flip=Uf, tile=2 if V.shape[0] == 16 else 4, carry=carry if ok else None)
                return dx_ if (ok or carry is None) else dx_ + carry
            if carry is not None and _carry_pitch(carry) is not None:
                return conv.conv2d_dgrad(dy, w, (dy.shape[2], dy.shape[3]), 1, 1, accum_into=carry)
            return _add_carry(conv.conv2d_dgrad(dy, w, (dy.shape[2], dy.shape[3]), 1, 1), carry)

        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and wino.dgrad_eligible(dy.shape):
            dx = _slot_deliver(slot, fused)
            dw = box[0]
        else:
            if ctx.needs_input_grad[0]:
                dx = _slot_deliver(slot, dgrad_only)
            if ctx.needs_input_grad[1]:
                dw = _side_run(lambda: wino.conv3x3_wgrad(V, dy, accum_into=gw), (V, dy)) if gw is not None else wino.conv3x3_wgrad(V, dy)
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = _bias_grad(dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1]), gb)
        return dx, dw, db, None, None


class _WinoConv3x3Levels(Function):
    """ONE 3x3 / stride 1 / pad 1 convolution (one filter, one bias) applied to several tensors -- detectron2's StandardRPNHead.conv
    over the FPN levels p2..p6 (configs/Base.yaml:49) -- with the Winograd-domain tiles of all of them side by side in one array: one
    batched GEMM per direction for all levels instead of one per level (the p4..p6 problems are 256 / 64 / 16 tile rows: alone each
    is a latency-bound launch), ONE weight-gradient problem whose row reduction sums the levels, and the transforms of every level
    reading / writing their row range of the shared arrays (csrc/winograd.hip, the *_rows entry points)."""

    @staticmethod
    def forward(ctx, w, bias, relu, *xs):
        ctx.set_materialize_grads(False)
        ctx.direct = (_direct_grad(w), _direct_grad(bias))
        ctx.slots = [_slot_enter(x, ctx.needs_input_grad[3 + i]) for i, x in enumerate(xs)]
        w_given = w
        xs, w = [_cl(x) for x in xs], _cl(w)
        shapes = [tuple(x.shape) for x in xs]
        tile = wino.levels_tile(shapes)
        assert tile in (2, 4), "conv3x3_levels_eligible() first"
        offs, rows = wino.level_rows(shapes, tile)
        need_flip = any(x.requires_grad for x in xs)
        key = (w.data_ptr(), tuple(w.shape), tile)
        cache = _wino_scope["cache"]
        U, Uf = cache.get(key, (None, None)) if cache is not None else (None, None)
        if U is None or (need_flip and Uf is None):
            U, Uf = wino.transform_weights(w, True, need_flip or Uf is not None, tile)
            if cache is not None:
                cache[key] = (U, Uf)
                if w is w_given:
                    _wino_scope["record"].append((w, tile, bool(need_flip or Uf is not None)))
        V = torch.empty(((tile + 2) ** 2, rows, w.shape[1]), dtype=torch.float32, device=w.device)
        for x, o in zip(xs, offs):
            wino.transform_input_rows(x, V, o, tile)
        Mt = wino.gemm_batched(V, U)
        ys = [wino.transform_output_rows(Mt, o, (sh[0], sh[2], sh[3]), bias, relu) for sh, o in zip(shapes, offs)]
        ctx.save_for_backward(V, w, Uf if need_flip else None, *(ys if relu else []))
        ctx.meta = (relu, bias is not None, tile, shapes, offs, rows)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        V, w, Uf, *ys = ctx.saved_tensors
        relu, has_bias, tile, shapes, offs, rows = ctx.meta
        K = w.shape[0]
        gw, gb = ctx.direct
        if gw is not None and not gw.is_contiguous(memory_format=CL):
            gw = None
        alloc = torch.zeros if any(d is None for d in dys) else torch.empty         # (a level without a gradient contributes zero rows)
        dM = alloc((V.shape[0], rows, K), dtype=torch.float32, device=V.device)
        Vd = alloc((V.shape[0], rows, K), dtype=torch.float32, device=V.device)
        db = None
        for l, dy in enumerate(dys):
            if dy is None:
                continue
            masked = relu and _relu_already_masked(dy)       # (the consumer's data-gradient kernel applied the mask)
            dy = _cl(dy)
            if relu and not masked:
                dy = bnpool.relu_bwd(dy.permute(0, 2, 3, 1), ys[l].permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            wino.transform_dy_in_rows(dy, dM, Vd, offs[l], tile)
            if has_bias and ctx.needs_input_grad[1]:
                part = _bias_grad(dy.permute(0, 2, 3, 1).reshape(-1, K), gb)
                if gb is None:
                    db = part if db is None else db + part
        dw = None
        if ctx.needs_input_grad[0]:
            if gw is not None:
                _side_run(lambda: wino.wgrad_into(V, dM, gw), (V, dM))
            else:
                dw = wino.transform_dweights(wino.gemm_batched_wgrad(V, dM), None)
        dxs = [None] * len(shapes)
        if any(ctx.needs_input_grad[3:]):
            if Uf is None:
                Uf = wino.transform_weights(w, want_u=False, want_flip=True, tile=tile)[1]
            Mx = wino.gemm_batched(Vd, Uf)
            for l, (sh, o) in enumerate(zip(shapes, offs)):
                if not ctx.needs_input_grad[3 + l]:
                    continue

                def dgrad(carry, sh=sh, o=o):
                    ok = carry is not None and _carry_pitch(carry) is not None
                    dx_ = wino.transform_output_rows(Mx, o, (sh[0], sh[2], sh[3]), carry=carry if ok else None)
                    return dx_ if (ok or carry is None) else dx_ + carry
                dxs[l] = _slot_deliver(ctx.slots[l], dgrad)
        return (dw, db, None, *dxs)


def conv3x3_levels_eligible(xs, w, stride=1, pad=1):
    """several tensors under one 3x3 filter through the shared Winograd arrays (_WinoConv3x3Levels)?"""
    if not (_WINOGRAD and len(xs) > 1 and w.shape[2] == 3 and w.shape[3] == 3 and stride == 1 and pad == 1):
        return False
    C, K = w.shape[1], w.shape[0]
    if C % 32 or K % 32 or C < 128 or K < 128 or any(x.dim() != 4 or x.shape[1] != C or x.dtype != torch.float32 for x in xs):
        return False
    return wino.levels_tile([tuple(x.shape) for x in xs]) != 0


def conv3x3_levels(xs, w, bias=None, relu=False):
    """[conv2d(x, w, bias, 1, 1, relu) for x in xs] with one GEMM per direction for all of them (see _WinoConv3x3Levels)"""
    return list(_WinoConv3x3Levels.apply(w, bias, bool(relu), *xs))


class _RPNHead16(Function):
    """objectness_logits + anchor_deltas of detectron2's StandardRPNHead over all FPN levels (csrc/rpn_head.hip): ts = the per-level
    ReLU outputs of the shared 3x3 convolution, (B, 256, H, W) CL -> per-level (B, 16, H, W) CL [3 logits | 12 deltas | 0].
    One launch forward, one for the data gradients (which come back already masked by the ReLU of the convolution below), two for the
    four parameter gradients, which land in the gradient bucket directly -- instead of, per level, a 16-wide implicit GEMM in each
    direction, a ReLU backward, a bias gradient and the adds that sum the per-level parameter gradients."""

    @staticmethod
    def forward(ctx, w_obj, b_obj, w_del, b_del, *ts):
        ctx.direct = (_direct_grad(w_obj), _direct_grad(b_obj), _direct_grad(w_del), _direct_grad(b_del))
        tn = [_cl(t).permute(0, 2, 3, 1) for t in ts]
        w_obj, w_del = _cl(w_obj), _cl(w_del)          # (K, 256, 1, 1) CL == (K, 256) row-major
        ys = det.head16_fwd(tn, w_obj, b_obj.contiguous(), w_del, b_del.contiguous())
        ctx.save_for_backward(w_obj, w_del, *tn)
        return tuple(y.permute(0, 3, 1, 2) for y in ys)

    @staticmethod
    def backward(ctx, *douts):
        w_obj, w_del, *tn = ctx.saved_tensors
        dys = [(_cl(d).permute(0, 2, 3, 1) if d is not None else torch.zeros(t.shape[:3] + (16,), dtype=torch.float32, device=t.device))
               for d, t in zip(douts, tn)]
        dts = [None] * len(tn)
        if any(ctx.needs_input_grad[4:]):
            dts = []
            for d in det.head16_dgrad(dys, tn, w_obj, w_del, relu_mask=True):
                d = d.permute(0, 3, 1, 2)
                d._omni_relu_masked = d._version           # read by _WinoConv3x3 / _Conv2d.backward: no second pass for the ReLU
                dts.append(d)
        grads = (None, None, None, None)
        if any(ctx.needs_input_grad[:4]):
            direct = ctx.direct
            if all(g is not None for g in direct) and direct[0].is_contiguous(memory_format=CL) and direct[2].is_contiguous(memory_format=CL):
                _side_run(lambda: det.head16_wgrad(dys, tn, accum_into=(direct[0].permute(0, 2, 3, 1), direct[1], direct[2].permute(0, 2, 3, 1),
                                                                           direct[3])), tuple(dys) + tuple(tn))
            else:
                gw_o, gb_o, gw_d, gb_d = det.head16_wgrad(dys, tn)
                grads = (gw_o.view(3, 1, 1, -1).permute(0, 3, 1, 2), gb_o, gw_d.view(12, 1, 1, -1).permute(0, 3, 1, 2), gb_d)
        return grads + tuple(dts)


def rpn_head16_eligible(ts, w_obj, w_del):
    return (all(t.shape[1] == det.HEAD16_C and t.dtype == torch.float32 for t in ts) and tuple(w_obj.shape[:2]) == (3, det.HEAD16_C)
            and tuple(w_del.shape[:2]) == (12, det.HEAD16_C) and tuple(w_obj.shape[2:]) == (1, 1) and len(ts) <= 8)


def rpn_head16(ts, w_obj, b_obj, w_del, b_del):
    """-> list of per-level (B, 16, H, W) CL head outputs"""
    return list(_RPNHead16.apply(w_obj, b_obj, w_del, b_del, *ts))


_wino_scope = {"cache": None, "record": [], "preloaded": None}     # (weight address, shape) -> (U, U'), only while a model forward is running


def wino_pretransform(owner, kind="_omni_wino_plan_train"):
    """every Winograd filter transform the next pass of `owner` will need (the plan its earlier passes left on it), as ONE launch
    outside the pass -> {(weight address, shape, tile): (U, U')}.  solver/graphed.py captures this as its own little graph and replays
    it on the idle weight-gradient stream beside the first layers of the forward pass (the transform moves 260 MB; it used to sit
    at the head of the critical path).  Hand the result to `wino_preloaded` around the pass that should use it."""
    plan = getattr(owner, kind, None) if _WINO_MULTI else None
    out = {}
    if not plan:
        return out
    items = [(w, True, flip, tile) for w, tile, flip in plan if w.is_contiguous(memory_format=CL)]
    for k in range(0, len(items), wino.WEIGHTS_MULTI_MAX):
        chunk = items[k:k + wino.WEIGHTS_MULTI_MAX]
        for (w, _, _, tile), uu in zip(chunk, wino.transform_weights_multi(chunk)):
            out[(w.data_ptr(), tuple(w.shape), tile)] = uu
    return out


class wino_preloaded:
    """with wino_preloaded(d): the passes inside take their filter transforms from `d` (wino_pretransform) instead of launching them"""

    def __init__(self, d):
        self.d = d

    def __enter__(self):
        self.prev, _wino_scope["preloaded"] = _wino_scope["preloaded"], self.d

    def __exit__(self, *a):
        _wino_scope["preloaded"] = self.prev
_WINO_MULTI = _os_environ_get("OMNI_WINO_WEIGHTS_MULTI", "1") != "0"          # A/B knob


class wino_weight_scope:
    """`with wino_weight_scope(owner):` around one forward pass.  A weight used by several convolutions of that pass (the RPN conv
    over the FPN levels) is transformed once; and the pass remembers on `owner` (the model) which filters it transformed, so the
    NEXT pass of the same kind (training / inference) transforms all of them with one launch at its start (wino.transform_weights_multi:
    the weights are fixed during a step) instead of one latency-bound launch in front of every convolution."""

    def __init__(self, owner=None):
        self.owner = owner

    def __enter__(self):
        self.prev = (_wino_scope["cache"], _wino_scope["record"])
        cache = _wino_scope["cache"] = {}
        _wino_scope["record"] = []
        self.kind = "_omni_wino_plan_train" if torch.is_grad_enabled() else "_omni_wino_plan_infer"
        plan = getattr(self.owner, self.kind, None) if (self.owner is not None and _WINO_MULTI) else None
        pre = _wino_scope["preloaded"]
        if pre is not None and self.kind == "_omni_wino_plan_train":
            cache.update(pre)
        elif plan:
            items = [(w, True, flip, tile) for w, tile, flip in plan if w.is_contiguous(memory_format=CL)]
            for k in range(0, len(items), wino.WEIGHTS_MULTI_MAX):
                chunk = items[k:k + wino.WEIGHTS_MULTI_MAX]
                for (w, _, _, tile), uu in zip(chunk, wino.transform_weights_multi(chunk)):
                    cache[(w.data_ptr(), tuple(w.shape), tile)] = uu

    def __exit__(self, *exc):
        record = _wino_scope["record"]
        if self.owner is not None and record and exc[0] is None:
            # filters this pass still had to transform one by one (first pass, or a shape the plan did not know): extend the plan
            plan = {(id(w), tile): (w, tile, flip) for w, tile, flip in (getattr(self.owner, self.kind, None) or [])}
            for w, tile, flip in record:
                old = plan.get((id(w), tile))
                plan[(id(w), tile)] = (w, tile, flip or (old[2] if old else False))
            try:
                object.__setattr__(self.owner, self.kind, list(plan.values()))     # (not a module attribute: keeps state_dict / children clean)
            except Exception:
                pass
        _wino_scope["cache"], _wino_scope["record"] = self.prev
        return False



_WINOGRAD = _os_environ_get("OMNI_WINOGRAD", "1") != "0"
# The RPN's shared 3x3 over all FPN levels through one GEMM per direction (conv3x3_levels).  MEASURED and left OFF for training
# (profiles/r04_ab_wino_levels.log: 11.33-11.35 ms with, 11.25-11.34 without -- the p4..p6 GEMMs it absorbs are paid back by 36-point
# transforms on the 16x16 / 8x8 maps and a third more rows in the p2 launch; inference: 623 against 616 images/s, but the smallest
# levels change from the direct kernel to the transform and one reference-written inference fixture flips a detection on it).
_WINO_LEVELS = _os_environ_get("OMNI_WINO_LEVELS", "0")               # "1" always | "0" never | "infer": eval mode only


def conv2d(x, w, bias=None, stride=1, pad=0, relu=False, want_stats=False):
    """want_stats: the caller is a conv -> BatchNorm pair in training mode; the result then carries `_omni_bn_partials`
    (per-workgroup sums / sums of squares written by the kernel that produced it) when the chosen kernel could emit them."""
    if _WINOGRAD and wino.eligible(x.shape, w.shape, stride, pad):
        y, parts = _WinoConv3x3.apply(x, w, bias, relu, want_stats)
    else:
        y, parts = _Conv2d.apply(x, w, bias, stride, pad, relu, want_stats)
    if parts.shape[0] > 0:
        y._omni_bn_partials = parts
    return y


class _GroupedConv2d(Function):
    @staticmethod
    def forward(ctx, x, w, groups, stride, pad):
        x, w = _cl(x), _cl(w)
        ctx.save_for_backward(x, w)
        ctx.cfg = (groups, stride, pad)
        return conv.grouped_conv2d_fwd(x, w, groups, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        groups, stride, pad = ctx.cfg
        dy = _cl(dy)
        dx = conv.grouped_conv2d_dgrad(dy, w, groups, (x.shape[2], x.shape[3]), stride, pad) if ctx.needs_input_grad[0] else None
        dw = conv.grouped_conv2d_wgrad(x, dy, groups, (w.shape[2], w.shape[3]), stride, pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


def grouped_conv2d(x, w, groups, stride=1, pad=0):
    """nn.Conv2d(C, K, k, stride, pad, groups=groups, bias=False): w (K, C / groups, k, k)"""
    return _GroupedConv2d.apply(x, w, groups, stride, pad)


class _DepthwiseConv2d(Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad):
        x = _cl(x)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad)
        return conv.dwconv_fwd(x, w, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = _cl(dy)
        dx = conv.dwconv_dgrad(dy, w, (x.shape[2], x.shape[3]), stride, pad) if ctx.needs_input_grad[0] else None
        dw = conv.dwconv_wgrad(x, dy, w.shape[2], stride, pad).contiguous() if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


def depthwise_conv2d(x, w, stride=1, pad=1):
    """nn.Conv2d(C, C, k, stride, pad, groups=C, bias=False): w (C, 1, k, k)"""
    return _DepthwiseConv2d.apply(x, w, stride, pad)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, bias, relu, w_grad_view=None, b_grad_view=None):
        gw = _direct_grad(w) if w_grad_view is None else w_grad_view
        ctx.direct = (gw if (gw is not None and gw.is_contiguous()) else None, _direct_grad(bias) if b_grad_view is None else b_grad_view)
        x, w = x.contiguous(), w.contiguous()
        y = conv.linear_fwd(x, w, bias, relu)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (relu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        relu, has_bias = ctx.cfg
        dy = dy.contiguous()
        if relu:
            dy = bnpool.relu_bwd(dy, y)
        gw, gb = ctx.direct
        dx = conv.linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _side_run(lambda: conv.linear_wgrad(x, dy, accum_into=gw), (x, dy)) if gw is not None else conv.linear_wgrad(x, dy)
        db = _bias_grad(dy, gb) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None


def linear(x, w, bias=None, relu=False, w_grad_view=None, b_grad_view=None):
    """w_grad_view / b_grad_view: contiguous views of the gradient bucket for `w` / `bias` when those are not parameters themselves
    (a view of one, FlattenLinear; the fused matrix of several, fused_linear); enables direct accumulation."""
    return _Linear.apply(x, w, bias, relu, w_grad_view, b_grad_view)


def fused_linear(x, weights, biases, rows, relu=False):
    """y = x . [W_0; W_1; ...; 0]^T + [b_0; b_1; ...; 0]: several nn.Linear evaluated as one GEMM of `rows` (>= sum of the members',
    zero padded) output columns.  When the optimizer laid the members out back to back (solver/build.py tag_fused_groups) the fused
    matrix, bias and both gradients are views of its buckets: no concatenation, one accumulating weight-gradient launch, no
    per-member gradient adds.  Otherwise the members are concatenated and autograd splits the gradient."""
    from .cubercnn.solver.build import fused_view
    cols = weights[0].shape[1]
    training = torch.is_grad_enabled() and any(w.requires_grad for w in weights)
    fw, fb = fused_view(weights, training), fused_view(biases, training)
    if fw is not None and fb is not None and fw[0].numel() == rows * cols and fb[0].numel() == rows:
        w, b = fw[0].view(rows, cols), fb[0]
        if training:
            return _Linear.apply(x, w.detach().requires_grad_(True), b.detach().requires_grad_(True), relu, fw[1].view(rows, cols), fb[1])
        return _Linear.apply(x, w, b, relu, None, None)
    pad = rows - sum(w.shape[0] for w in weights)
    w = torch.cat(list(weights) + ([weights[0].new_zeros(pad, cols)] if pad else []), dim=0)
    b = torch.cat(list(biases) + ([biases[0].new_zeros(pad)] if pad else []), dim=0)
    return _Linear.apply(x, w, b, relu, None, None)


def _wino_weights(w, tile, need_flip):
    """(U, U') of filter `w` through the pass's cache (wino_weight_scope), recording a miss for the next pass's batched launch"""
    key = (w.data_ptr(), tuple(w.shape), tile)
    cache = _wino_scope["cache"]
    U, Uf = cache.get(key, (None, None)) if cache is not None else (None, None)
    if U is None or (need_flip and Uf is None):
        U, Uf = wino.transform_weights(w, True, need_flip or Uf is not None, tile)
        if cache is not None:
            cache[key] = (U, Uf)
            _wino_scope["record"].append((w, tile, bool(need_flip or Uf is not None)))
    return U, Uf


class _BNReLUWinoConv(Function):
    """conv -> BatchNorm -> ReLU -> 3x3 convolution with the normalised activation never materialised (the inside of every DLA /
    torchvision BasicBlock, /root/reference/cubercnn/modeling/backbone/dla.py:60-66): x is the RAW output of the first convolution with
    its epilogue statistics; BatchNorm's finalize turns them into (scale, shift), and the Winograd input transform of the second
    convolution applies scale / shift / ReLU while it loads its tiles (omni_wino_in_affine).  Saves the write and the read of the
    normalised tensor and one launch per block; the backward pass is the two layers' usual kernels (the BatchNorm's ReLU mask is
    recomputed from x, so nothing needed the normalised tensor anyway)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, partials, w, want_stats):
        ctx.set_materialize_grads(False)
        gg, gb = _direct_grad(gamma), _direct_grad(beta)
        ctx.bn_direct = (gg, gb) if (gg is not None and gb is not None) else None
        ctx.w_direct = _direct_grad(w)
        x, w = _cl(x), _cl(w)
        mean_rstd, scale_shift = bnpool.bn_finalize_fwd(x, gamma, beta, running_mean, running_var, partials, eps, momentum)
        tile = wino.tile_size(x.shape)
        need_flip = wino.dgrad_eligible(x.shape)
        U, Uf = _wino_weights(w, tile, need_flip)
        parts = None
        if want_stats:
            y, V, parts = wino.conv3x3_fwd(x, w, None, False, U=U, tile=tile, want_stats=True, in_affine=scale_shift, in_relu=True)
        else:
            y, V = wino.conv3x3_fwd(x, w, None, False, U=U, tile=tile, in_affine=scale_shift, in_relu=True)
        ctx.save_for_backward(x, gamma, mean_rstd, scale_shift, V, w, Uf if need_flip else None)
        parts = _parts_out(parts, y)
        ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, dy, _parts_grad=None):
        x, gamma, mean_rstd, scale_shift, V, w, Uf = ctx.saved_tensors
        dy = _cl(dy)
        gw = ctx.w_direct
        if gw is not None and not gw.is_contiguous(memory_format=CL):
            gw = None
        # the second convolution: data gradient (= the gradient of the normalised activation) and weight gradient
        if wino.dgrad_eligible(dy.shape):
            dmid, dw = wino.conv3x3_backward(V, dy, w, Uf, accum_into=gw, side_run=_side_run)
        else:
            dmid = conv.conv2d_dgrad(dy, w, (dy.shape[2], dy.shape[3]), 1, 1)
            dw = _side_run(lambda: wino.conv3x3_wgrad(V, dy, accum_into=gw), (V, dy)) if gw is not None else wino.conv3x3_wgrad(V, dy)
        # the BatchNorm + ReLU: mask recomputed from x and (scale, shift)
        dx, _, dgamma, dbeta = bnpool.bn_bwd(x, dmid, None, gamma, mean_rstd, True, want_dres=False, accum_into=ctx.bn_direct,
                                             scale_shift=scale_shift)
        return dx, dgamma, dbeta, None, None, None, None, None, dw, None


# OFF by default: measured SLOWER on the DLA-34 step (11.93 / 11.95 ms with, 11.89 without; profiles/r03_ab_bn_wino_fuse.log): the input
# transform reads every activation 2.25 times (overlapping 6x6 windows) and now normalises it 2.25 times, with 8 more live registers
# next to its 144-register tile -- that costs more than the 5 us bn_apply launch and the 2 x 4-17 MB it saves
_BN_WINO_FUSE = _os_environ_get("OMNI_BN_WINO_FUSE", "0") != "0"      # A/B knob
_BN_WINO_FUSE_MAX_PIX = int(_os_environ_get("OMNI_BN_WINO_FUSE_MAX_PIX", str(1 << 30)))     # ... only on maps of at most this many pixels


def bn_relu_conv3x3(x, bn, conv_mod, want_stats):
    """`conv_mod(bn(x, relu=True))` for a 3x3 / stride 1 / pad 1 / bias-free `conv_mod` (layers.Conv2d) behind a training-mode
    BatchNorm `bn` (layers.BatchNorm2d) whose input `x` carries its producer's epilogue statistics: fused when the convolution takes
    the Winograd path, the plain two-module sequence otherwise."""
    w = conv_mod.weight
    parts = getattr(x, "_omni_bn_partials", None)
    if (_BN_WINO_FUSE and _WINOGRAD and parts is not None and bn.training and torch.is_grad_enabled() and conv_mod.bias is None
            and conv_mod.stride[0] == 1 and conv_mod.padding[0] == 1 and w.requires_grad and x.requires_grad
            and wino.eligible(x.shape, w.shape, 1, 1) and _BN_REMASK and w.is_contiguous(memory_format=CL)
            and x.shape[0] * x.shape[2] * x.shape[3] <= _BN_WINO_FUSE_MAX_PIX):
        y, p2 = _BNReLUWinoConv.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                      bn.running_var if bn.track_running_stats else None, bn.eps, bn.momentum, parts, w, bool(want_stats))
        if bn.track_running_stats and bn.num_batches_tracked is not None and not bn.defer_counter:
            bn.num_batches_tracked += 1
        if p2.shape[0] > 0:
            y._omni_bn_partials = p2
        return y
    return conv_mod(bn(x, relu=True))


_BN_REMASK = _os_environ_get("OMNI_BN_REMASK", "1") != "0"     # A/B knob
# backward reductions from the data-gradient transform above the layer (wino.transform_output_bn_bwd): the F(4x4) output transform that
# writes a BatchNorm's output gradient also emits that layer's (sum dz, sum dz * xhat) partial rows, so its backward is ONE launch
# (finalize folded into the apply pass, csrc/bn_pool.hip) instead of a reduction pass + apply.  Round 3 measured it neutral with the
# separate finalize launch (317.9 / 316.2 images/s with, 316.3 without); round 5, without that launch: 10.905 -> 10.877 ms per step
# (profiles/r05_ab_bn_fuse.log) -- on by default, OMNI_BN_BWD_FUSE=0 restores the reduction pass.
_BN_BWD_FUSE = _os_environ_get("OMNI_BN_BWD_FUSE", "1") != "0"


class _BatchNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, relu, eps, momentum, partials=None):
        gg, gb = _direct_grad(gamma), _direct_grad(beta)
        ctx.direct = (gg, gb) if (gg is not None and gb is not None) else None
        ctx.res_slot = _slot_enter(residual, residual is not None and ctx.needs_input_grad[5])
        x = _cl(x)
        res = _cl(residual) if residual is not None else None
        y, mean_rstd, scale_shift = bnpool.bn_fwd(x, gamma, beta, running_mean, running_var, res, relu, eps, momentum, partials)
        # ReLU mask for the backward pass: the output y, or (no residual) the 2C-float (scale, shift) pair -- y > 0 is then
        # recomputed from x, which the backward kernels read anyway, and y is not read again (bnpool.bn_bwd)
        remask = relu and residual is None and _BN_REMASK
        ctx.save_for_backward(x, gamma, mean_rstd, (y if relu and not remask else None), (scale_shift if remask else None))
        ctx.cfg = (relu, residual is not None)
        if _BN_BWD_FUSE and residual is None and (remask or not relu):
            # what the convolution that consumes y needs to leave this layer's backward reductions behind (_WinoConv3x3.backward)
            y._omni_bn_below = (x, mean_rstd, scale_shift if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean_rstd, y, scale_shift = ctx.saved_tensors
        relu, has_res = ctx.cfg
        parts = getattr(dy, "_omni_bn_bwd_parts", None)
        if parts is not None:       # made for THIS layer (same statistics tensor), and dy reached us unchanged
            parts = parts[0] if parts[1].data_ptr() == mean_rstd.data_ptr() and dy.is_contiguous(memory_format=CL) else None
        want_dres = has_res and ctx.needs_input_grad[5]
        box = []

        # (dy that is a channel slice of the Root's concatenated gradient is read where it lies)
        dyk = dy if (parts is None and not dy.is_contiguous(memory_format=CL) and _carry_pitch(dy) is not None) else _cl(dy)

        def run(carry):     # carry: what the other consumers of the residual tensor contributed, added where dres is written
            ok = carry is not None and parts is None and _carry_pitch(carry) is not None
            dx_, dres_, dgamma_, dbeta_ = bnpool.bn_bwd(x, dyk, y, gamma, mean_rstd, relu, want_dres=want_dres, accum_into=ctx.direct,
                                                        scale_shift=scale_shift, partials=parts, res_carry=carry if ok else None)
            box.append((dx_, dgamma_, dbeta_))
            return dres_ if (ok or carry is None) else dres_ + carry

        dres = _slot_deliver(ctx.res_slot, run) if want_dres else run(None)
        dx, dgamma, dbeta = box[0]
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, residual=None, relu=False, eps=1e-5, momentum=0.1):
    """statistics: taken from `x._omni_bn_partials` when the producer of x emitted them (functional.conv2d(want_stats=True))"""
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, residual, relu, eps, momentum, getattr(x, "_omni_bn_partials", None))


class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.slot = _slot_enter(x, ctx.needs_input_grad[0])
        x = _cl(x)
        ctx.save_for_backward(x)
        return bnpool.maxpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors

        dyk = dy if (not dy.is_contiguous(memory_format=CL) and _carry_pitch(dy) is not None) else _cl(dy)

        def run(carry):
            if carry is not None and (_carry_pitch(carry) is None or (x.shape[2] | x.shape[3]) & 1):
                return bnpool.maxpool2_bwd(x, dyk) + carry
            return bnpool.maxpool2_bwd(x, dyk, carry=carry)
        return _slot_deliver(ctx.slot, run)


class _AvgPool2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        ctx.hw = (x.shape[2], x.shape[3])
        return bnpool.avgpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return bnpool.avgpool2_bwd(_cl(dy), ctx.hw)


class _Subsample2(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.slot = _slot_enter(x, ctx.needs_input_grad[0])
        x = _cl(x)
        ctx.hw = (x.shape[2], x.shape[3])
        return bnpool.subsample2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        def run(carry):
            if carry is not None and _carry_pitch(carry) is None:
                return bnpool.subsample2_bwd(_cl(dy), ctx.hw) + carry
            return bnpool.subsample2_bwd(_cl(dy), ctx.hw, carry=carry)
        return _slot_deliver(ctx.slot, run)


class _Upsample2Add(Function):
    @staticmethod
    def forward(ctx, lat, top):
        ctx.slot = _slot_enter(top, ctx.needs_input_grad[1])
        return bnpool.upsample2_add(_cl(lat), _cl(top))

    @staticmethod
    def backward(ctx, dout):
        dout = _cl(dout)

        def run(carry):
            if carry is not None and _carry_pitch(carry) is None:
                return bnpool.upsample2_bwd(dout) + carry
            return bnpool.upsample2_bwd(dout, carry=carry)
        return dout, _slot_deliver(ctx.slot, run)


class _CatChannels(Function):
    """torch.cat(xs, dim=1) of the DLA Root (dla.py:171) whose backward hands each input its channel slice of the gradient THROUGH
    the input's fan-in slot: the slice -- a strided view, nothing is copied -- becomes the carry the input's other consumers add
    their gradients to."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.slots = [_slot_enter(x, ctx.needs_input_grad[i]) for i, x in enumerate(xs)]
        ctx.sizes = [x.shape[1] for x in xs]
        return torch.cat(xs, 1)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for i, (slot, c) in enumerate(zip(ctx.slots, ctx.sizes)):
            piece = g[:, off:off + c]
            off += c
            outs.append(_slot_deliver(slot, lambda carry, piece=piece: _add_carry(piece, carry)) if ctx.needs_input_grad[i] else None)
        return tuple(outs)


class _CatConv1x1(Function):
    """conv1x1(torch.cat(xs, 1)) of the DLA Root (dla.py:166-172) without the concatenated copy in the forward pass: the kernel reads
    every reduction slab from the child that holds its channels (csrc/conv_gemm.hip, ConvP::xs), with the BatchNorm statistics in
    its epilogue like any conv -> BatchNorm pair.  Backward: ONE data gradient of the concatenated input whose channel slices --
    strided views -- go to the children through their fan-in slots exactly as `_CatChannels` hands them over; the weight gradient
    reads the children in place as well.  Bit-identical to `_Conv2d(_CatChannels(xs))`."""

    @staticmethod
    def forward(ctx, w, want_stats, *xs):
        ctx.set_materialize_grads(False)
        ctx.direct = _direct_grad(w)
        ctx.slots = [_slot_enter(x, ctx.needs_input_grad[2 + i]) for i, x in enumerate(xs)]
        xs = [_cl(x) for x in xs]
        w = _cl(w)
        y, parts = conv.conv1x1_multi_fwd(xs, w, want_stats=want_stats)
        ctx.save_for_backward(w, *xs)
        parts = _parts_out(parts, y)
        ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, dy, _parts_grad=None):
        w, *xs = ctx.saved_tensors
        dy = _cl(dy)
        gw = ctx.direct
        if gw is not None and not gw.is_contiguous(memory_format=CL):
            gw = None
        outs = [None] * len(xs)
        if any(ctx.needs_input_grad[2:]):
            g = conv.conv2d_dgrad(dy, w, (dy.shape[2], dy.shape[3]), 1, 0)
            off = 0
            for i, (slot, x) in enumerate(zip(ctx.slots, xs)):
                piece = g[:, off:off + x.shape[1]]
                off += x.shape[1]
                if ctx.needs_input_grad[2 + i]:
                    outs[i] = _slot_deliver(slot, lambda carry, piece=piece: _add_carry(piece, carry))
        dw = None
        if ctx.needs_input_grad[0]:
            def wgrad():
                return conv.conv1x1_multi_wgrad(xs, dy, accum_into=gw)
            dw = _side_run(wgrad, (*xs, dy)) if gw is not None else wgrad()
        return (dw, None) + tuple(outs)


def cat_conv1x1(xs, w, want_stats=False):
    """conv2d(torch.cat(xs, 1), w) for a 1 x 1, stride-1, bias-free convolution (see `_CatConv1x1`); the caller checked
    `conv.multi_src_eligible(xs, w)`"""
    y, parts = _CatConv1x1.apply(w, want_stats, *xs)
    if parts.shape[0] > 0:
        y._omni_bn_partials = parts
    return y


class _PadInputChannels(Function):
    """(K, C, R, S) filter -> (K, C4, R, S) with zero input channels appended: the 3-channel stem filter against the image padded to
    4 channels (dla.py:241-245).  The padded copy lives in a persistent buffer whose extra channels stay zero, so a step pays one
    strided copy instead of a zero-fill + concatenation, and the backward is a view (round 5)."""

    @staticmethod
    def forward(ctx, w, c4, holder):
        import weakref
        buf = holder.get("buf")
        if buf is None or buf.shape[0] != w.shape[0] or buf.shape[1] != c4 or buf.device != w.device or buf.dtype != w.dtype:
            buf = holder["buf"] = torch.zeros((w.shape[0], c4, w.shape[2], w.shape[3]), dtype=w.dtype, device=w.device).contiguous(memory_format=CL)
        last = holder.get("last")
        if torch.is_grad_enabled() and last is not None and last() is not None and not torch.cuda.is_current_stream_capturing():
            # the alias handed out by an EARLIER grad-mode forward is still alive -- a convolution saved it for a backward that has not
            # run yet (forward, forward, backward, backward; two model calls under one loss).  Writing the buffer now would bump its
            # version under that saved tensor: this call gets a padded tensor of its own instead.  Captured steps never
            # take this branch: a capture owns one forward and one backward.
            out = torch.zeros_like(buf)
            out[:, : w.shape[1]].copy_(w)
            ctx.c = w.shape[1]
            return out
        buf[:, : w.shape[1]].copy_(w)
        ctx.c = w.shape[1]
        out = buf.detach()           # (a fresh alias per call: the buffer object itself never carries an autograd node)
        if torch.is_grad_enabled():
            holder["last"] = weakref.ref(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, : ctx.c], None, None


def pad_input_channels(w, c4, holder):
    return _PadInputChannels.apply(w, c4, holder)


def cat_channels(xs):
    return _CatChannels.apply(*xs)


def max_pool2(x):
    return _MaxPool2.apply(x)


def avg_pool2(x):
    """nn.AvgPool2d(kernel_size=2, stride=2)"""
    return _AvgPool2.apply(x)


def subsample2(x):
    return _Subsample2.apply(x)


def upsample2_add(lat, top):
    return _Upsample2Add.apply(lat, top)


class _ROIAlign(Function):
    """feats: logical (B,C,H,W) CL tensors of the FPN levels -> (R, C, P, P) CL (physically (R,P,P,C))."""

    @staticmethod
    def forward(ctx, rois, batch_idx, levels, scales, P, *feats):
        ctx.slots = [_slot_enter(f, ctx.needs_input_grad[5 + i]) for i, f in enumerate(feats)]
        feats = [_cl(f) for f in feats]
        nhwc = [f.permute(0, 2, 3, 1) for f in feats]
        out = det.roi_align_fwd(nhwc, scales, rois, batch_idx, levels, P)
        ctx.save_for_backward(rois, batch_idx, levels)
        ctx.meta = (scales, P, [tuple(f.shape) for f in nhwc])
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        rois, batch_idx, levels = ctx.saved_tensors
        scales, P, shapes = ctx.meta
        if det.roi_align_bwd_deterministic(P, rois.shape[0], shapes[0][3]):
            dfe = [torch.empty(s, dtype=torch.float32, device=dout.device) for s in shapes]      # every element written exactly once
            det.roi_align_bwd_det(dfe, scales, rois, batch_idx, levels, P, _cl(dout).permute(0, 2, 3, 1).contiguous())
        else:
            dfe = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in shapes]
            det.roi_align_bwd(dfe, scales, rois, batch_idx, levels, P, _cl(dout).permute(0, 2, 3, 1))
        return (None, None, None, None, None) + tuple(_slot_deliver(slot, lambda carry, d=d: _add_carry(d.permute(0, 3, 1, 2), carry))
                                                      for slot, d in zip(ctx.slots, dfe))


def roi_align(feats, scales, rois, batch_idx, levels, P):
    return _ROIAlign.apply(rois, batch_idx, levels, tuple(scales), P, *feats)


class _ROIAlignLegacy(Function):
    """POOLER_TYPE "ROIAlign" (torchvision roi_align with aligned=False: no half-pixel shift, ROI sides >= 1 pixel), round 6: the general
    forward kernel and its atomic backward with the switch exposed (omni_roi_align_fwd_mode / _bwd_mode)."""

    @staticmethod
    def forward(ctx, rois, batch_idx, levels, scales, P, *feats):
        ctx.slots = [_slot_enter(f, ctx.needs_input_grad[5 + i]) for i, f in enumerate(feats)]
        feats = [_cl(f) for f in feats]
        nhwc = [f.permute(0, 2, 3, 1) for f in feats]
        out = det.roi_align_fwd_mode(nhwc, scales, rois, batch_idx, levels, P, False)
        ctx.save_for_backward(rois, batch_idx, levels)
        ctx.meta = (scales, P, [tuple(f.shape) for f in nhwc])
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        rois, batch_idx, levels = ctx.saved_tensors
        scales, P, shapes = ctx.meta
        dfe = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in shapes]
        det.roi_align_bwd_mode(dfe, scales, rois, batch_idx, levels, P, False, _cl(dout).permute(0, 2, 3, 1).contiguous())
        return (None, None, None, None, None) + tuple(_slot_deliver(slot, lambda carry, d=d: _add_carry(d.permute(0, 3, 1, 2), carry))
                                                      for slot, d in zip(ctx.slots, dfe))


def roi_align_legacy(feats, scales, rois, batch_idx, levels, P):
    return _ROIAlignLegacy.apply(rois, batch_idx, levels, tuple(scales), P, *feats)


class _ROIPool(Function):
    """POOLER_TYPE "ROIPool" (torchvision roi_pool: whole-pixel ROI corners, maximum over each bin, gradient to the argmax pixel),
    round 6: omni_roi_pool_fwd / _bwd (csrc/roi_align.hip)."""

    @staticmethod
    def forward(ctx, rois, batch_idx, levels, scales, P, *feats):
        ctx.slots = [_slot_enter(f, ctx.needs_input_grad[5 + i]) for i, f in enumerate(feats)]
        feats = [_cl(f) for f in feats]
        nhwc = [f.permute(0, 2, 3, 1) for f in feats]
        out, arg = det.roi_pool_fwd(nhwc, scales, rois, batch_idx, levels, P)
        ctx.save_for_backward(batch_idx, levels, arg)
        ctx.meta = (scales, P, [tuple(f.shape) for f in nhwc])
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        batch_idx, levels, arg = ctx.saved_tensors
        scales, P, shapes = ctx.meta
        dfe = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in shapes]
        det.roi_pool_bwd(dfe, scales, batch_idx, levels, P, _cl(dout).permute(0, 2, 3, 1).contiguous(), arg)
        return (None, None, None, None, None) + tuple(_slot_deliver(slot, lambda carry, d=d: _add_carry(d.permute(0, 3, 1, 2), carry))
                                                      for slot, d in zip(ctx.slots, dfe))


def roi_pool(feats, scales, rois, batch_idx, levels, P):
    return _ROIPool.apply(rois, batch_idx, levels, tuple(scales), P, *feats)


class _ROIAlignShared(Function):
    """Training: the cube head pools the SAME sampled boxes as the box head with an identical pooler (the reference builds
    two ROIPoolers with the same resolution / sampling ratio / type, roi_heads.py:166-171, and calls them on the same
    `proposal_boxes`, :267 and :362), restricted to the first `first` slots of each image's block of `per_image` ROIs.
    One ROIAlign forward serves both heads -> (x_all, x_first); backward folds d(x_first) into d(x_all) and runs ONE
    ROIAlign backward (one zero-fill + one atomic pass per level instead of two, and no autograd adds of the per-level
    feature gradients)."""

    @staticmethod
    def forward(ctx, rois, batch_idx, levels, scales, P, per_image, first, *feats):
        ctx.slots = [_slot_enter(f, ctx.needs_input_grad[7 + i]) for i, f in enumerate(feats)]
        feats = [_cl(f) for f in feats]
        nhwc = [f.permute(0, 2, 3, 1) for f in feats]
        out, sub = det.roi_align_fwd2(nhwc, scales, rois, batch_idx, levels, P, per_image, first)   # (R, P, P, C), (R/per_image*first, P, P, C)
        ctx.save_for_backward(rois, batch_idx, levels)
        ctx.meta = (scales, P, per_image, first, [tuple(f.shape) for f in nhwc])
        return out.permute(0, 3, 1, 2), sub.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, d_all, d_first):
        rois, batch_idx, levels = ctx.saved_tensors
        scales, P, per_image, first, shapes = ctx.meta
        d = _cl(d_all).permute(0, 2, 3, 1) if d_all is not None else None
        df = _cl(d_first).permute(0, 2, 3, 1) if d_first is not None else None
        ref = d if d is not None else df
        # the five level gradients are views of ONE zero-filled slab (one fill launch), and the kernel adds the cube head's gradient
        # to the box head's on the fly (round 3: a 103 MB clone of d, a strided add and five fills before)
        sizes = [s[0] * s[1] * s[2] * s[3] for s in shapes]
        ordered = det.roi_align_bwd_deterministic(P, rois.shape[0], shapes[0][3])
        slab = (torch.empty if ordered else torch.zeros)(sum(sizes), dtype=torch.float32, device=ref.device)
        dfe, off = [], 0
        for shp, n in zip(shapes, sizes):
            dfe.append(slab[off:off + n].view(shp))
            off += n
        if ordered:
            # owner-computes backward: one wave per 8 x 8 tile adds the ROIs' contributions in ROI order and writes each element once
            det.roi_align_bwd_det(dfe, scales, rois, batch_idx, levels, P, d.contiguous() if d is not None else None,
                                  dout2=df.contiguous() if df is not None else None, per_image=per_image, first=first)
        elif df is not None and P == 7:
            det.roi_align_bwd(dfe, scales, rois, batch_idx, levels, P, d, dout2=df.contiguous(), per_image=per_image, first=first)
        else:
            if df is not None:            # (general pooler resolution: merge on the host side as before)
                if d is None:
                    d = torch.zeros((rois.shape[0], P, P, df.shape[3]), dtype=torch.float32, device=df.device)
                else:
                    d = d.clone()
                C = d.shape[3]
                d.view(-1, per_image, P, P, C)[:, :first] += df.reshape(-1, first, P, P, C)
            det.roi_align_bwd(dfe, scales, rois, batch_idx, levels, P, d)
        # (each level's gradient goes through the feature's fan-in slot: the RPN head's data gradient, which runs later, reads it
        # in its output transform instead of an add kernel over the whole map)
        return (None,) * 7 + tuple(_slot_deliver(slot, lambda carry, t=t: _add_carry(t.permute(0, 3, 1, 2), carry))
                                   for slot, t in zip(ctx.slots, dfe))


def roi_align_shared(feats, scales, rois, batch_idx, levels, P, per_image, first):
    return _ROIAlignShared.apply(rois, batch_idx, levels, tuple(scales), P, per_image, first, *feats)


def _scalar(t):
    return t.reshape(1).contiguous().float()


_coef_cache = {}


def _coef(values, device, dtype=torch.float32):
    """small constant vector on the device, built once (before any hipGraph capture: the warm-up steps run first)"""
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _coef_cache.get(key)
    if t is None:
        t = _coef_cache[key] = torch.tensor(key[0], dtype=dtype, device=device)
    return t


def _scalar_grads(g, n):
    """the n one-element gradient tensors a loss kernel reads, from the gradient of its n-vector WITHOUT a copy launch where the
    vector's gradient is one broadcast scalar (stride 0: what `_SumVectors` and autograd's own sum hand down)"""
    if g.dtype != torch.float32:
        g = g.float()
    if g.dim() == 1 and g.stride(0) == 0:
        one = g.as_strided((1,), (1,))
        return (one,) * n
    g = g.contiguous()
    return tuple(g[k:k + 1] for k in range(n))


class LossDict(dict):
    """The reference's loss dictionary (name -> 0-d tensor) that also remembers the small vectors its entries are views of.
    `sum(d.values())` (tools/train_net.py:180) works as always; `total_loss(d)` computes the same sum from the vectors with
    two launches instead of one add per entry in the forward and a zero-fill + copy + accumulate per entry in the backward."""

    def __init__(self, *a, vectors=None, **kw):
        super().__init__(*a, **kw)
        self.vectors = list(vectors or [])          # [(vector tensor, names it contributes to this dict)]

    def update(self, other=(), **kw):
        super().update(other, **kw)
        if isinstance(other, LossDict):
            self.vectors += other.vectors


class _SumVectors(Function):
    """sum of all elements of a few small vectors.  Its backward hands every vector the upstream scalar as a stride-0 view -- no
    launch; the loss functions read it as the one scalar it is (`_scalar_grads`).  The forward value comes from `sums3[k]`, one
    element of the (3,) result of ONE glue.sum_vectors launch that serves both partial sums of a staged step and their total
    (round 6: cat + sum per partial sum and an add for the total before)."""

    @staticmethod
    def forward(ctx, sums3, k, *vecs):
        ctx.shapes = [tuple(v.shape) for v in vecs]
        return sums3[k]

    @staticmethod
    def backward(ctx, g):
        # (stride-0 views in the SHAPE of every input: a vector that is not 1-D gets a gradient of its own shape)
        g = g.reshape(())
        return (None, None) + tuple(g.expand(sh) for sh in ctx.shapes)


def _vec_ok(vecs):
    return 0 < len(vecs) <= glue.MAXV and all(v.dtype == torch.float32 and v.is_contiguous() and v.numel() <= 4096 for v in vecs)


def sum_vectors(vecs):
    """scalar sum of a list of small loss vectors (differentiable)"""
    return sum_vectors2(vecs, ())[0]


def sum_vectors2(first, rest):
    """-> (sum of the vectors in `first`, sum of those in `rest` or None, total) with ONE launch; the two partial sums are
    differentiable (separately: they are the roots of different backward stages, solver/graphed.py), the total is a plain value"""
    first, rest = list(first), list(rest)
    if not _vec_ok(first + rest):      # (not the loss vectors of this package: plain torch)
        a = torch.cat([v.reshape(-1) for v in first]).sum()
        b = torch.cat([v.reshape(-1) for v in rest]).sum() if rest else None
        return a, b, (a.detach() + b.detach()) if rest else a.detach()
    sums3 = glue.sum_vectors(first, rest)
    a = _SumVectors.apply(sums3, 0, *first)
    b = _SumVectors.apply(sums3, 1, *rest) if rest else None
    return a, b, sums3[2]


def total_loss(losses):
    """== sum(losses.values()) (up to fp32 summation order)"""
    vecs = getattr(losses, "vectors", None)
    if vecs and sorted(n for _, names in vecs for n in names) == sorted(losses.keys()):
        return sum_vectors([v for v, _ in vecs])
    return sum(losses.values())


class _RPNLoss(Function):
    """-> ((2,) = [sum BCE*t, sum L1*t] * inv_norm * loss weights, raw sums); level tensors are the fused head outputs (B,16,H,W) CL."""

    @staticmethod
    def forward(ctx, anchors, labels, matched_idx, gt, gt_off, inv_norm, weights, plain, *levels):
        ctx.set_materialize_grads(False)      # no zero tensors for the non-differentiable side outputs
        with forked():                        # (behind the labelling, beside the proposal selection: see set_branch_stream)
            lv = [_cl(t).permute(0, 2, 3, 1) for t in levels]
            pack = det.LevelPack(lv)
            sums = det.rpn_loss_fwd(pack, anchors, labels, matched_idx, gt, gt_off, plain)
            # (fp64 sums x fp64 coefficients, rounded once into the fp32 loss vector: one launch)
            vec = glue.scale_vec(sums, (inv_norm * weights[0], inv_norm * weights[1]), n=2)
        ctx.pack = pack
        ctx.save_for_backward(anchors, labels, matched_idx, gt, gt_off)
        ctx.inv_norm, ctx.weights, ctx.plain = inv_norm, weights, plain
        ctx.mark_non_differentiable(sums)
        return vec, sums

    @staticmethod
    def backward(ctx, g, _):
        anchors, labels, matched_idx, gt, gt_off = ctx.saved_tensors
        if ctx.weights != (1.0, 1.0):
            g = g.float() * _coef(ctx.weights, g.device)
        g_cls, g_loc = _scalar_grads(g, 2)
        grads = det.rpn_loss_bwd(ctx.pack, anchors, labels, matched_idx, gt, gt_off, g_cls, g_loc, ctx.inv_norm, ctx.plain)
        return (None,) * 8 + tuple(t.permute(0, 3, 1, 2) for t in grads)


def rpn_loss(levels, anchors, labels, matched_idx, gt, gt_off, inv_norm, weights=(1.0, 1.0), plain=False):
    """plain: OBJECTNESS_UNCERTAINTY 'none' instead of 'IoUness' -> ((2,) [cls, loc] weighted losses, raw sums)"""
    return _RPNLoss.apply(anchors, labels, matched_idx, gt, gt_off, inv_norm, tuple(float(w) for w in weights), bool(plain), *levels)


class _BoxLoss(Function):
    """-> ((2,) = [loss_cls, loss_box_reg] * loss weights, raw sums)"""

    @staticmethod
    def forward(ctx, pred, K, cls, prop, gt, gt_row, weights, loss_w):
        ctx.set_materialize_grads(False)      # no zero tensors for the non-differentiable side outputs
        pred = pred.contiguous()
        sums = det.box_loss_fwd(pred, K, cls, prop, gt, gt_row, weights)
        ctx.save_for_backward(pred, cls, prop, gt, gt_row, sums)
        ctx.meta = (K, weights, loss_w)
        ctx.mark_non_differentiable(sums)
        # (fp64 sums / max(fp64 count, 1) x loss weights, rounded once into the fp32 loss vector: one launch, round 6)
        vec = glue.scale_vec(sums, loss_w, denom=sums[2:3], denom_min=1.0, n=2)
        return vec, sums

    @staticmethod
    def backward(ctx, g, _):
        pred, cls, prop, gt, gt_row, sums = ctx.saved_tensors
        K, weights, loss_w = ctx.meta
        if loss_w != (1.0, 1.0):
            g = g.float() * _coef(loss_w, g.device)
        g_cls, g_reg = _scalar_grads(g, 2)
        return (det.box_loss_bwd(pred, K, cls, prop, gt, gt_row, sums, g_cls, g_reg, weights),) + (None,) * 7


def box_loss(pred, K, cls, prop, gt, gt_row, weights=(10.0, 10.0, 5.0, 5.0), loss_w=(1.0, 1.0)):
    """-> ((2,) [loss_cls, loss_box_reg] weighted, raw sums)"""
    return _BoxLoss.apply(pred, K, cls, prop, gt, gt_row, tuple(weights), tuple(float(w) for w in loss_w))


class _CubeLoss(Function):
    """-> ((6,) = [loss_dims, loss_xy, loss_z, loss_pose, loss_joint, uncert] * coef ; red (24) raw reductions + logging stats)."""

    @staticmethod
    def forward(ctx, head, K, boxes, cls, img, Ks, v2r, priors, gt3d, gtpose, gt_row, loss_w, mode, coef, clusters):
        ctx.set_materialize_grads(False)      # no zero tensors for the non-differentiable side outputs
        head = head.contiguous()
        vals, jac, red = det.cube_loss_fwd(head, K, boxes, cls, img, Ks, v2r, priors, gt3d, gtpose, gt_row, loss_w, mode, clusters)
        ctx.save_for_backward(vals, jac, red, cls, boxes)
        ctx.meta = (head.shape[0], K, head.shape[1], mode, coef, clusters)
        ctx.mark_non_differentiable(red)
        return glue.scale_vec(red, coef, n=6), red

    @staticmethod
    def backward(ctx, g, _):
        vals, jac, red, cls, boxes = ctx.saved_tensors
        F_, K, ldh, mode, coef, clusters = ctx.meta
        g = g.float()
        if g.dim() == 1 and g.stride(0) in (0, 1):
            gk = glue.scale_vec(g, coef, n=6)              # (a broadcast scalar x 6 coefficients -> a contiguous 6-vector: one launch)
        else:
            gk = (g * _coef(coef, g.device)).contiguous()
        dhead = det.cube_loss_bwd(vals, jac, red, gk, cls, boxes, F_, K, ldh, mode, clusters)
        return (dhead,) + (None,) * 14


def cube_loss(head, K, boxes, cls, img, Ks, v2r, priors, gt3d, gtpose, gt_row, loss_w=(1.0, 1.0, 1.0, 1.0, 1.0), mode=det.CUBE_MODE_BASE,
              coef=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), clusters=None):
    """coef: what each of [loss_dims, loss_xy, loss_z, loss_pose, loss_joint, uncert] is multiplied by on the way out;
    clusters: None or (bins, zscales (K, bins), zstats (K, bins, 2) | None) for CLUSTER_BINS > 1
    -> ((6,) weighted losses, red (24))"""
    return _CubeLoss.apply(head, K, boxes, cls, img, Ks, v2r, priors, gt3d, gtpose, gt_row, tuple(loss_w), int(mode),
                           tuple(float(c) for c in coef), clusters)


class _MaxPool3s2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        ctx.save_for_backward(x)
        return bnpool.maxpool3s2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return bnpool.maxpool3s2_bwd(x, _cl(dy))


def max_pool3s2(x):
    """nn.MaxPool2d(3, stride=2, padding=1)"""
    return _MaxPool3s2.apply(x)
