"""Convolution / linear launchers over the implicit-GEMM kernels (csrc/conv_gemm.hip).

Tensors cross this boundary in torch's channels_last memory format: logically (N,C,H,W) /
(K,C,R,S) like the reference's nn.Conv2d, physically NHWC / KRSC, so `permute(0,2,3,1)` is a
free contiguous view.  Channel counts must be multiples of 4 (16-byte loads); callers pad the
3-channel image and the odd-sized prediction heads.
"""
import os

import torch

from .. import lib as _lib
from . import detmode as _det


def _nhwc(t):
    """logical NCHW channels_last tensor -> (N,H,W,C) contiguous view (no copy when already CL)."""
    if t.dim() != 4:
        raise ValueError("expected a 4-D tensor")
    v = t.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        v = v.contiguous()
    return v


def to_channels_last(t):
    return t.contiguous(memory_format=torch.channels_last)


def _fwd_launch(L, x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, relu, tile, splits, like, stats=None, stats_rows=0, nblk_addr=None):
    """omni_conv2d_fwd_algo / _stats, or -- deterministic mode -- omni_conv2d_fwd_det with the workspace its own plan asks for"""
    st = _lib.stream_of(like)
    if not _det.on():
        if stats is not None:
            L.call("omni_conv2d_fwd_stats", x, w, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, _lib.ptr(stats), stats_rows, nblk_addr, st)
        else:
            L.call("omni_conv2d_fwd_algo", x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, int(relu), tile, splits, st)
        return
    plan, addr = _det.new_plan()
    L.call("omni_conv2d_fwd_det", x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, int(relu), tile, splits, None, 0, None, None, 0,
           None, 0, addr, st)
    ws, wsf, ctr, nctr = _det.workspace(like, plan)
    L.call("omni_conv2d_fwd_det", x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, int(relu), int(plan[0]), int(plan[1]),
           _lib.ptr(stats), stats_rows, nblk_addr, _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, st)


def _dgrad_launch(L, dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, tile, splits, like):
    st = _lib.stream_of(like)
    if not _det.on():
        L.call("omni_conv2d_dgrad_algo", dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, tile, splits, st)
        return
    plan, addr = _det.new_plan()
    L.call("omni_conv2d_dgrad_det", dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, tile, splits, None, 0, None, 0, addr, st)
    ws, wsf, ctr, nctr = _det.workspace(like, plan)
    L.call("omni_conv2d_dgrad_det", dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, int(plan[0]), int(plan[1]),
           _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, st)


def _wgrad_launch(L, x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, tile, like):
    st = _lib.stream_of(like)
    if not _det.on():
        L.call("omni_conv2d_wgrad_algo", x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, tile, st)
        return
    plan, addr = _det.new_plan()
    L.call("omni_conv2d_wgrad_det", x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, tile, None, 0, None, 0, addr, st)
    ws, wsf, ctr, nctr = _det.workspace(like, plan)
    # (the tile is passed back as the caller gave it: `tile` also carries the workgroup-order request, + 16 / + 32)
    L.call("omni_conv2d_wgrad_det", x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, tile, _lib.ptr(ws), wsf, _lib.ptr(ctr),
           nctr, None, st)


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, relu=False, tile=0, splits=0):
    """x (N,C,H,W) CL, w (K,C,R,S) CL -> y (N,K,OH,OW) CL.  tile/splits: explicit algorithm (0 = the launcher's choice)."""
    xv, wv = _nhwc(x), _nhwc(w)
    N, H, W, C = xv.shape
    K, R, S, C2 = wv.shape
    assert C == C2, (C, C2)
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    L = _lib.check_device(xv, wv, bias)
    out = torch.empty((N, OH, OW, K), dtype=torch.float32, device=x.device)
    _fwd_launch(L, _lib.ptr(xv), _lib.ptr(wv), _lib.ptr(bias), _lib.ptr(out), N, H, W, C, K, R, S, stride, pad, C, K, relu, tile, splits, x)
    return out.permute(0, 3, 1, 2)


STATS_ROWS = 4096       # capacity (partial rows) of the BatchNorm statistics buffer a producing kernel may fill


def _stats_buf(K, device):
    return torch.empty((STATS_ROWS, 2 * K), dtype=torch.float32, device=device)


def _nblk_cell():
    import ctypes
    cell = ctypes.c_int(0)
    return cell, ctypes.addressof(cell)


def conv2d_fwd_stats(x, w, stride=1, pad=0):
    """conv2d_fwd (no bias, no ReLU) + BatchNorm partial statistics from the epilogue -> (y, partial (nblk, 2K) or None)"""
    xv, wv = _nhwc(x), _nhwc(w)
    N, H, W, C = xv.shape
    K, R, S, _ = wv.shape
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    L = _lib.check_device(xv, wv)
    out = torch.empty((N, OH, OW, K), dtype=torch.float32, device=x.device)
    stats = _stats_buf(K, x.device)
    cell, addr = _nblk_cell()
    _fwd_launch(L, _lib.ptr(xv), _lib.ptr(wv), None, _lib.ptr(out), N, H, W, C, K, R, S, stride, pad, C, K, False, 0, 0, x, stats=stats,
                stats_rows=STATS_ROWS, nblk_addr=addr)
    return out.permute(0, 3, 1, 2), (stats[:cell.value] if cell.value > 0 else None)


MULTI_SRC_MAX = 6


def multi_src_eligible(xs, w):
    """omni_conv2d_fwd_multi_det serves: a 1 x 1 filter over 2..6 same-sized maps whose channel counts are multiples of 32"""
    if not (2 <= len(xs) <= MULTI_SRC_MAX) or w.dim() != 4 or w.shape[2] != 1 or w.shape[3] != 1:
        return False
    n, _, h, wd = xs[0].shape
    if any(x.dim() != 4 or x.shape[0] != n or x.shape[2] != h or x.shape[3] != wd or x.shape[1] % 32 or x.dtype != torch.float32 for x in xs):
        return False
    return sum(x.shape[1] for x in xs) == w.shape[1] and n * h * wd * max(x.shape[1] for x in xs) < (1 << 31)


def conv1x1_multi_fwd(xs, w, bias=None, relu=False, want_stats=False, tile=0, splits=0):
    """conv2d_fwd(torch.cat(xs, 1), w) for a 1 x 1 filter WITHOUT the concatenated copy (the DLA Root, dla.py:166-172): every
    reduction slab reads its channels from the map that holds them; bit-identical to the single-tensor call.
    -> (y (N,K,H,W) CL, BatchNorm partial statistics (nblk, 2K) or None)"""
    import ctypes
    xv = [_nhwc(x) for x in xs]
    wv = _nhwc(w)
    N, H, W, _ = xv[0].shape
    K = wv.shape[0]
    L = _lib.check_device(*xv, wv, bias)
    out = torch.empty((N, H, W, K), dtype=torch.float32, device=xv[0].device)
    n = len(xv)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in xv])
    cs = (ctypes.c_int * n)(*[int(t.shape[3]) for t in xv])
    pa, ca = ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p)
    stats = _stats_buf(K, out.device) if want_stats and bias is None and not relu else None
    cell, addr = _nblk_cell()
    st = _lib.stream_of(xv[0])
    head = (pa, ca, n, _lib.ptr(wv), _lib.ptr(bias), _lib.ptr(out), N, H, W, K, K, int(relu))
    if not _det.on():
        L.call("omni_conv2d_fwd_multi_det", *head, tile, splits, _lib.ptr(stats), STATS_ROWS if stats is not None else 0, addr, None, 0, None, 0,
               None, st)
    else:
        plan, paddr = _det.new_plan()
        L.call("omni_conv2d_fwd_multi_det", *head, tile, splits, None, 0, None, None, 0, None, 0, paddr, st)
        ws, wsf, ctr, nctr = _det.workspace(xv[0], plan)
        L.call("omni_conv2d_fwd_multi_det", *head, int(plan[0]), int(plan[1]), _lib.ptr(stats), STATS_ROWS if stats is not None else 0, addr,
               _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, st)
    return out.permute(0, 3, 1, 2), (stats[:cell.value] if stats is not None and cell.value > 0 else None)


def stem_conv_fwd_stats(x, w):
    xv, wv = _nhwc(x), _nhwc(w)
    N, H, W, C = xv.shape
    K, R = wv.shape[0], wv.shape[1]
    L = _lib.check_device(xv, wv)
    out = torch.empty((N, H, W, K), dtype=torch.float32, device=x.device)
    stats = _stats_buf(K, x.device)
    cell, addr = _nblk_cell()
    L.call("omni_stem_conv_fwd_stats", _lib.ptr(xv), _lib.ptr(wv), _lib.ptr(out), N, H, W, C, K, R, C, K, _lib.ptr(stats), STATS_ROWS, addr,
           _lib.stream_of(x))
    return out.permute(0, 3, 1, 2), (stats[:cell.value] if cell.value > 0 else None)


_S2_DGRAD = os.environ.get("OMNI_S2_DGRAD", "1") != "0"            # A/B knob: 0 = the generic kernel's four grid.z parity classes
_S2_DGRAD_MIN_WGS = int(os.environ.get("OMNI_S2_DGRAD_MIN_WGS", "192"))
# the 32-channel form (DLA level 2's entry, 32 -> 64 at 256 x 256): 35 us alone against the generic kernel's 56; INSIDE the step its own
# row reads 128 us against 84 (profiles/r06_trace_table_final.txt: end of backward, beside the busiest stretch of the weight-gradient
# stream) and the STEP is still the shorter one with it: 10.753 / 10.756 ms against 10.772 / 10.766 without (profiles/r06_ab_s2_dgrad_c32.log)
# -- what it leaves to the other stream counts too.  OMNI_S2_DGRAD_C32=0 switches this form off.
_S2_DGRAD_C32 = os.environ.get("OMNI_S2_DGRAD_C32", "1") != "0"


def s2_dgrad_eligible(N, H, W, C, K, R, S, stride, pad):
    """3x3 / stride 2 / pad 1 with channel counts in multiples of 32, and enough 8 x 8 dy tiles (x 64-channel groups) to fill the chip:
    the small-map layers (DLA level 4 / 5 entries at 512 x 512 input: 64 and 16 tiles) stay on the split-K form of the generic kernel"""
    if not _S2_DGRAD or (R, S, stride, pad) != (3, 3, 2, 1) or (C % 32) or (K % 32):
        return False
    if C % 64 and not _S2_DGRAD_C32:
        return False
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    wgs = N * ((OH + 7) // 8) * ((OW + 7) // 8) * (C // 64 if C % 64 == 0 else C // 32)
    return wgs >= _S2_DGRAD_MIN_WGS


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=0, tile=0, splits=0, accum_into=None):
    """dy (N,K,OH,OW) CL, w (K,C,R,S) CL -> dx (N,C,H,W) CL.
    accum_into: gradient fan-in target (logical (N,C,H,W), NHWC memory with any pixel pitch -- functional._carry_pitch): the result
    is ADDED to it in place (split reductions: atomics on top of its content, no zero-fill) and it is returned."""
    dyv, wv = _nhwc(dy), _nhwc(w)
    N, OH, OW, K = dyv.shape
    K2, R, S, C = wv.shape
    assert K == K2
    H, W = in_hw
    L = _lib.check_device(dyv, wv)
    if tile == 0 and splits == 0 and s2_dgrad_eligible(N, H, W, C, K, R, S, stride, pad):
        # round 6: all four parity classes of a dx tile from one staged dy tile (csrc/dgrad_s2.hip)
        if accum_into is not None:
            assert tuple(accum_into.shape) == (N, C, H, W) and accum_into.stride(1) == 1
            L.call("omni_conv2d_s2_dgrad", _lib.ptr(dyv), _lib.ptr(wv), accum_into.data_ptr(), N, H, W, C, K, K, accum_into.stride(3), 1,
                   _lib.stream_of(dy))
            return accum_into
        dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
        L.call("omni_conv2d_s2_dgrad", _lib.ptr(dyv), _lib.ptr(wv), _lib.ptr(dx), N, H, W, C, K, K, C, 0, _lib.stream_of(dy))
        return dx.permute(0, 3, 1, 2)
    if accum_into is not None:
        assert tuple(accum_into.shape) == (N, C, H, W) and accum_into.stride(1) == 1
        _dgrad_launch(L, _lib.ptr(dyv), _lib.ptr(wv), accum_into.data_ptr(), N, H, W, C, K, R, S, stride, pad, K, accum_into.stride(3), 1,
                      tile, splits, dy)
        return accum_into
    dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    _dgrad_launch(L, _lib.ptr(dyv), _lib.ptr(wv), _lib.ptr(dx), N, H, W, C, K, R, S, stride, pad, K, C, 0, tile, splits, dy)
    return dx.permute(0, 3, 1, 2)


def conv2d_wgrad(x, dy, ksize, stride=1, pad=0, accum_into=None, tile=0):
    """x (N,C,H,W) CL, dy (N,K,OH,OW) CL -> dw (K,C,R,S) CL.  accum_into: a (K,C,R,S) CL tensor (e.g. the
    parameter's view of the flat gradient bucket) that the result is atomically added to (returns None)."""
    xv, dyv = _nhwc(x), _nhwc(dy)
    N, H, W, C = xv.shape
    K = dyv.shape[3]
    R, S = ksize
    L = _lib.check_device(xv, dyv)
    if accum_into is not None:
        tgt = accum_into.permute(0, 2, 3, 1)
        assert tgt.is_contiguous() and tgt.shape == (K, R, S, C)
        if tile == 0 and R == S and _defer_direct((xv, dyv, tgt, N, H, W, C, K, R, stride, pad, None)):
            return None           # (inside wino.batched_wgrads(): leaves with the stage's other direct weight gradients in one launch)
        _wgrad_launch(L, _lib.ptr(xv), _lib.ptr(dyv), _lib.ptr(tgt), N, H, W, C, K, R, S, stride, pad, C, K, 1, tile, x)
        return None
    dw = torch.empty((K, R, S, C), dtype=torch.float32, device=x.device)
    _wgrad_launch(L, _lib.ptr(xv), _lib.ptr(dyv), _lib.ptr(dw), N, H, W, C, K, R, S, stride, pad, C, K, 0, tile, x)
    return dw.permute(0, 3, 1, 2)


# The stage's direct weight gradients in ONE launch.  MEASURED and left OFF (OMNI_WGRAD_BATCH=1 enables it):
# 10.85 / 10.84 ms with it against 10.83 / 10.80 ms without on one box (profiles/r06_ab_wgrad_batch.log) -- unlike the Winograd-domain
# GEMMs (0.23 -> 0.66 of peak from the same move in round 4) these launches are not short of workgroups: each already spreads ~1024
# of them over a split reduction, and what holds them at 0.25 of peak is the operand path of the 128 x 64 tile (PMC: MFMA-busy 0.26),
# which a shared launch does not change.  Results are bit-identical either way (tests/test_conv.py::test_wgrad_batch_*).
WGRAD_BATCH = os.environ.get("OMNI_WGRAD_BATCH", "0") == "1"
_direct_deferred = None        # list while wino.batched_wgrads() is open (the weight-gradient stream's capture of one backward stage)


def _defer_direct(item):
    """queue an accumulating direct weight gradient (x, dy, dw view, N, H, W, C, K, R, stride, pad, sources or None) for the
    stage's multi-problem launch; False = not inside the context (or switched off): the caller launches it itself"""
    if _direct_deferred is None or not WGRAD_BATCH or not _det.on():
        return False
    _direct_deferred.append(item)
    return True


def open_direct_batch():
    global _direct_deferred
    prev, _direct_deferred = _direct_deferred, []
    return prev


def close_direct_batch(prev, launch=True):
    """launch what was queued (omni_conv2d_wgrad_batch_det) and restore the enclosing state"""
    global _direct_deferred
    items, _direct_deferred = _direct_deferred, prev
    if launch and items:
        conv2d_wgrad_batch(items)


def conv2d_wgrad_batch(items):
    """items: [(x NHWC, dy NHWC, dw KRSC view to ADD into, N, H, W, C, K, R, stride, pad, [source NHWC tensors] or None)] -> every dw
    += its weight gradient, in one launch per tile shape (csrc/conv_gemm.hip conv_wgrad_multi_kernel); bit-identical to
    conv2d_wgrad(accum_into=...) / conv1x1_multi_wgrad(accum_into=...) on each.  Two items that add into the same view never share a
    launch (their order would be the hardware's): the second one starts a new batch."""
    import ctypes
    batch, seen, rest = [], set(), []
    for it in items:
        (rest if it[2].data_ptr() in seen else batch).append(it)
        seen.add(it[2].data_ptr())
    n = len(batch)
    L = _lib.check_device(*[t for it in batch for t in it[:3]])
    P = ctypes.c_void_p
    arr = lambda vals: ctypes.cast((P * len(vals))(*vals), P)                     # noqa: E731
    ints = lambda vals: ctypes.cast((ctypes.c_int * len(vals))(*[int(v) for v in vals]), P)         # noqa: E731
    xs_flat, cs_flat, nsrc = [], [], []
    for it in batch:
        src = it[11] or []
        nsrc.append(len(src))
        xs_flat += [t.data_ptr() for t in src] + [0] * (6 - len(src))
        cs_flat += [int(t.shape[3]) for t in src] + [0] * (6 - len(src))
    head = (arr([it[0].data_ptr() for it in batch]), arr([it[1].data_ptr() for it in batch]), arr([it[2].data_ptr() for it in batch]),
            ints([it[3] for it in batch]), ints([it[4] for it in batch]), ints([it[5] for it in batch]), ints([it[6] for it in batch]),
            ints([it[7] for it in batch]), ints([it[8] for it in batch]), ints([it[9] for it in batch]), ints([it[10] for it in batch]),
            ints([1] * n), arr(xs_flat), ints(cs_flat), ints(nsrc), n)
    like = batch[0][1]
    st = _lib.stream_of(like)
    plan, paddr = _det.new_plan()
    L.call("omni_conv2d_wgrad_batch_det", *head, None, 0, None, 0, paddr, st)
    ws, wsf, ctr, nctr = _det.workspace(like, plan)
    L.call("omni_conv2d_wgrad_batch_det", *head, _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, st)
    if rest:
        conv2d_wgrad_batch(rest)


def conv1x1_multi_wgrad(xs, dy, accum_into=None, tile=0):
    """conv2d_wgrad(torch.cat(xs, 1), dy, (1, 1)) without the concatenated copy (omni_conv2d_wgrad_multi_det); same contract"""
    import ctypes
    xv = [_nhwc(x) for x in xs]
    dyv = _nhwc(dy)
    N, H, W, K = dyv.shape
    C = sum(int(t.shape[3]) for t in xv)
    L = _lib.check_device(*xv, dyv)
    n = len(xv)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in xv])
    cs = (ctypes.c_int * n)(*[int(t.shape[3]) for t in xv])
    pa, ca = ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p)
    if accum_into is not None:
        tgt = accum_into.permute(0, 2, 3, 1)
        assert tgt.is_contiguous() and tgt.shape == (K, 1, 1, C)
        if tile == 0 and n <= 6 and _defer_direct((xv[0], dyv, tgt, N, H, W, C, K, 1, 1, 0, xv)):
            return None
        dw, acc = tgt, 1
    else:
        dw, acc = torch.empty((K, 1, 1, C), dtype=torch.float32, device=dyv.device), 0
    st = _lib.stream_of(dyv)
    head = (pa, ca, n, _lib.ptr(dyv), _lib.ptr(dw), N, H, W, K, K, acc, tile)
    if not _det.on():
        L.call("omni_conv2d_wgrad_multi_det", *head, None, 0, None, 0, None, st)
    else:
        plan, paddr = _det.new_plan()
        L.call("omni_conv2d_wgrad_multi_det", *head, None, 0, None, 0, paddr, st)
        ws, wsf, ctr, nctr = _det.workspace(dyv, plan)
        L.call("omni_conv2d_wgrad_multi_det", *head, _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, st)
    return None if accum_into is not None else dw.permute(0, 3, 1, 2)


def _engine_eligible(M, C, K):
    """The round-2 GEMM engine (csrc/gemm_engine.hip) wins where one workgroup per CU gets a long reduction to stream: the
    fc1-class layers (2048 x 12544 -> 1024: 124 vs 102 TFLOP/s forward, tools/bench_engine.py).  Everything shorter or narrower
    stays on the 64x64 / 128x128 tile kernels, whose 2-4 workgroups per CU hide the per-tile prologue better."""
    return M >= 1024 and C >= 4096 and K >= 512 and (C % 32) == 0


def linear_fwd(x, w, bias=None, relu=False):
    """x (M, Cin), w (Cout, Cin) -> (M, Cout): the 1x1 / H=W=1 case of the same kernel."""
    M, C = x.shape
    K = w.shape[0]
    L = _lib.check_device(x, w, bias)
    if _engine_eligible(M, C, K):
        from . import gemm as _gemm
        if relu and _det.on():      # ordered split: the last-arriving workgroup holds the complete sum, bias + ReLU there
            return _gemm.gemm(x, w, _gemm.NT, bias=bias, relu=True, tile=2, splits=2)
        if relu:            # split reduction ends in atomics: bias rides on split 0, the ReLU needs the complete sum
            out = _gemm.gemm(x, w, _gemm.NT, bias=bias, tile=2, splits=2)
            return out.clamp_(min=0)
        return _gemm.gemm(x, w, _gemm.NT, bias=bias, tile=2, splits=2)
    out = torch.empty((M, K), dtype=torch.float32, device=x.device)
    _fwd_launch(L, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), M, 1, 1, C, K, 1, 1, 1, 0, C, K, relu, 0, 0, x)
    return out


_DGRAD_NT = os.environ.get("OMNI_FC_DGRAD_NT", "1") != "0"
_DGRAD_FORM = os.environ.get("OMNI_FC_DGRAD_FORM", "nn")            # "nn": the engine reads W as it is | "nt": transpose W first (rounds 2-3)
_FC_BALANCED = os.environ.get("OMNI_FC_BALANCED", "1") != "0"
# Round 5: the fc1-class WEIGHT gradient is back on the tile kernel.  It runs on the weight-gradient stream beside the critical path, and
# the engine's balanced 128 x 128 form compiles to 256 VGPRs + 154 AGPRs = 410 registers per lane (one workgroup per CU because of its
# 96 KB of LDS, so hipcc budgets the whole file): while its 256 persistent workgroups hold every CU, no wave that needs more than ~100
# registers can start anywhere on the chip, and the main stream's next kernels (wino4_out: 254 VGPRs) waited out its 550 us
# (gpurun_out/r05a_timeline.txt; VERDICT r4 weak 3 saw the same stall).  The tile kernel (152 registers, 49 KB of LDS, non-persistent)
# is slower in isolation and shares the CUs: 10.85 -> 10.72 ms / step (profiles/r05_ab_fc1_wgrad.log; engine on 224 workgroups: 10.80).
# OMNI_FC_WGRAD_ENGINE_MIN_ROWS=1024 restores the engine (A/B knob).
_FC_WGRAD_ENGINE_MIN_ROWS = int(os.environ.get("OMNI_FC_WGRAD_ENGINE_MIN_ROWS", str(1 << 30)))
_FC_WGRAD_WGS = int(os.environ.get("OMNI_FC_WGRAD_WGS", "0"))              # persistent workgroups of the engine form (0 = one per CU)
_FC_WGRAD_SPLITS = int(os.environ.get("OMNI_FC_WGRAD_SPLITS", "-1"))      # -1 = balanced (gemm.BALANCED) | 1 = whole tiles only


def linear_dgrad(dy, w):
    M, K = dy.shape
    C = w.shape[1]
    L = _lib.check_device(dy, w)
    if _DGRAD_NT and M >= 512 and C >= 4096 and K >= 512 and (K % 32) == 0:
        # fc1-class data gradient dX = dY W on the LDS-DMA engine.  Round 4: its NN form reads W as it is (box head 494 us, cube head 160 us,
        # same sums bit for bit); rounds 2-3 transposed W first (51 MB, two launches per step on the critical path) for the NT main
        # loop: 540 / 175 us including the transpose (profiles/r04_fc1_nn.log).  OMNI_FC_DGRAD_FORM=nt: the old form
        from . import gemm as _gemm
        if _DGRAD_FORM == "nt":
            return _gemm.gemm(dy, _gemm.transpose2d(w), _gemm.NT, tile=2, splits=_gemm.BALANCED if _FC_BALANCED else 1)
        return _gemm.gemm(dy, w, _gemm.NN, tile=2, splits=_gemm.BALANCED if _FC_BALANCED else 1)
    dx = torch.empty((M, C), dtype=torch.float32, device=dy.device)
    _dgrad_launch(L, _lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), M, 1, 1, C, K, 1, 1, 1, 0, K, C, 0, 0, 0, dy)
    return dx


def linear_wgrad(x, dy, accum_into=None):
    M, C = x.shape
    K = dy.shape[1]
    L = _lib.check_device(x, dy)
    if accum_into is not None:
        assert accum_into.is_contiguous() and accum_into.shape == (K, C)
        if _FC_BALANCED and M >= _FC_WGRAD_ENGINE_MIN_ROWS and C >= 4096 and K >= 512 and (K % 4) == 0 and (C % 4) == 0:
            # fc1-class weight gradient dW += dY^T X on the LDS-DMA engine's TN form with the balanced work split: 784 tiles of
            # 128 x 128 are 3.06 rounds of 256 workgroups -- the plain launch pays 4 (csrc/gemm_engine.hip, BAL)
            from . import gemm as _gemm
            _gemm.gemm(dy, x, _gemm.TN, out=accum_into, accumulate=True, tile=2, splits=_FC_WGRAD_SPLITS, workgroups=_FC_WGRAD_WGS)
            return None
        _wgrad_launch(L, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(accum_into), M, 1, 1, C, K, 1, 1, 1, 0, C, K, 1, 0, x)
        return None
    dw = torch.empty((K, C), dtype=torch.float32, device=x.device)
    _wgrad_launch(L, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), M, 1, 1, C, K, 1, 1, 1, 0, C, K, 0, 0, x)
    return dw


def stem_eligible(x_shape, w_shape, stride, pad):
    """the full-resolution few-channel layers served by csrc/stem_conv.hip: (C, R) = (4, 7) or (16, 3), 16 outputs, stride 1"""
    K, C, R, S = w_shape
    return K == 16 and R == S and stride == 1 and pad == R // 2 and (C, R) in ((4, 7), (16, 3)) and x_shape[1] == C


def stem_conv_fwd(x, w):
    """x (N,C,H,W) CL, w (16,C,R,R) CL -> (N,16,H,W) CL"""
    xv, wv = _nhwc(x), _nhwc(w)
    N, H, W, C = xv.shape
    K, R = wv.shape[0], wv.shape[1]
    L = _lib.check_device(xv, wv)
    out = torch.empty((N, H, W, K), dtype=torch.float32, device=x.device)
    L.call("omni_stem_conv_fwd", _lib.ptr(xv), _lib.ptr(wv), _lib.ptr(out), N, H, W, C, K, R, C, K, _lib.stream_of(x))
    return out.permute(0, 3, 1, 2)


def stem_first_eligible(x_shape, w):
    """the first layer on the 4-channel padded image with its 3-channel filter as the model holds it (round 6): 7x7, 16 outputs,
    KRSC memory, deterministic reductions on (the narrow filter gradient is written by the ordered finalize launch)"""
    return (_det.on() and tuple(w.shape) == (16, 3, 7, 7) and x_shape[1] == 4 and w.is_contiguous(memory_format=torch.channels_last)
            and w.dtype == torch.float32)


def stem_first_fwd(x, w, want_stats):
    """x (N,4,H,W) CL (4th channel zero), w (16,3,7,7) CL -> (y (N,16,H,W) CL, BatchNorm partial rows or None)"""
    xv = _nhwc(x)
    wv = w.permute(0, 2, 3, 1)
    assert wv.is_contiguous()
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, wv)
    out = torch.empty((N, H, W, 16), dtype=torch.float32, device=x.device)
    stats = _stats_buf(16, x.device) if want_stats else None
    cell, addr = _nblk_cell()
    L.call("omni_stem_conv_fwd_cw", _lib.ptr(xv), _lib.ptr(wv), 3, _lib.ptr(out), N, H, W, C, 16, 7, C, 16, _lib.ptr(stats),
           STATS_ROWS if want_stats else 0, addr, _lib.stream_of(x))
    return out.permute(0, 3, 1, 2), (stats[:cell.value] if want_stats and cell.value > 0 else None)


def stem_first_wgrad(x, dy, accum_into=None):
    """-> dw (16,3,7,7) CL, or None after ADDING it into accum_into (the parameter's KRSC gradient view)"""
    xv, dv = _nhwc(x), _nhwc(dy)
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, dv)
    dst = accum_into.permute(0, 2, 3, 1) if accum_into is not None else torch.empty((16, 7, 7, 3), dtype=torch.float32, device=x.device)
    assert dst.is_contiguous() and tuple(dst.shape) == (16, 7, 7, 3)
    acc = int(accum_into is not None)
    plan, addr = _det.new_plan()
    L.call("omni_stem_conv_wgrad_det_cw", _lib.ptr(xv), _lib.ptr(dv), _lib.ptr(dst), 3, N, H, W, C, 16, 7, C, 16, acc, None, 0, addr, _lib.stream_of(x))
    ws = torch.empty(max(int(plan[3]), 1), dtype=torch.float32, device=x.device)
    L.call("omni_stem_conv_wgrad_det_cw", _lib.ptr(xv), _lib.ptr(dv), _lib.ptr(dst), 3, N, H, W, C, 16, 7, C, 16, acc, _lib.ptr(ws), int(plan[3]), None,
           _lib.stream_of(x))
    return None if accum_into is not None else dst.permute(0, 3, 1, 2)


_STEM_DGRAD = os.environ.get("OMNI_STEM_DGRAD", "1") != "0"        # A/B: 0 = the round-4 data gradients of level0 / level1


def stem_dgrad_eligible(x_shape, w_shape, stride, pad):
    """data gradients served by csrc/stem_conv.hip (round 5): 3x3 16 -> 16 stride 1 (level0, filter rotated inside the kernel) and
    3x3 stride 2 pad 1 16 -> 32 (level1, parity classes inside one launch)"""
    K, C, R, S = w_shape
    if not _STEM_DGRAD or R != 3 or S != 3 or pad != 1 or C != 16 or x_shape[1] != 16:
        return False
    return (stride == 1 and K == 16) or (stride == 2 and K == 32)


def stem_conv_dgrad(dy, w, in_hw, stride):
    """dy (N,K,OH,OW) CL, w (K,16,3,3) CL (the layer's forward filter) -> dx (N,16,H,W) CL"""
    dyv, wv = _nhwc(dy), _nhwc(w)
    N, OH, OW, K = dyv.shape
    H, W = in_hw
    assert (OH, OW) == ((H - 1) // stride + 1, (W - 1) // stride + 1) and tuple(wv.shape) == (K, 3, 3, 16)
    L = _lib.check_device(dyv, wv)
    dx = torch.empty((N, H, W, 16), dtype=torch.float32, device=dy.device)
    L.call("omni_stem_conv_dgrad" if stride == 1 else "omni_stem_conv_s2_dgrad", _lib.ptr(dyv), _lib.ptr(wv), _lib.ptr(dx), N, H, W, 16, K, 3,
           K, 16, _lib.stream_of(dy))
    return dx.permute(0, 3, 1, 2)


_STEM_WGRAD_ALL = os.environ.get("OMNI_STEM_WGRAD_ALL", "1") != "0"       # A/B: 0 = only the 16 -> 16 stride-1 layer (rounds 2-3)


def stem_wgrad_eligible(x_shape, w_shape, stride, pad):
    """weight gradients served by stem_conv_wgrad_kernel: the two stride-1 stem layers and (16 -> 32, 3x3, stride 2) = DLA-34 level1.
    The 4-channel 7x7 form only with the deterministic reductions on: its 3136-element filter gradient met 3072 waves' worth of
    fp32 atomics (0.26 ms against the implicit GEMM's 0.21), the partial rows + fixed-order finalize do not"""
    K, C, R, S = w_shape
    if R != S or pad != R // 2 or x_shape[1] != C:
        return False
    if stride == 1:
        return K == 16 and ((C, R) == (16, 3) or ((C, R) == (4, 7) and _det.on() and _STEM_WGRAD_ALL))
    return _STEM_WGRAD_ALL and stride == 2 and (K, C, R) == (32, 16, 3) and x_shape[2] % 2 == 0 and x_shape[3] % 2 == 0


def stem_conv_wgrad(x, dy, R, accum_into=None, stride=1):
    """x (N,C,H,W) CL, dy (N,K,H/stride,W/stride) CL -> dw (K,C,R,R) CL; accum_into: KRSC-contiguous gradient view to ADD into."""
    xv, dv = _nhwc(x), _nhwc(dy)
    N, H, W, C = xv.shape
    K = dv.shape[3]
    L = _lib.check_device(xv, dv)
    on = _det.on()

    def call(dst, acc, ws, wsf, plan):
        if stride == 2:
            L.call("omni_stem_conv_s2_wgrad", _lib.ptr(xv), _lib.ptr(dv), _lib.ptr(dst), N, H, W, C, K, R, C, K, acc, int(on), ws, wsf, plan,
                   _lib.stream_of(x))
        elif on:
            L.call("omni_stem_conv_wgrad_det", _lib.ptr(xv), _lib.ptr(dv), _lib.ptr(dst), N, H, W, C, K, R, C, K, acc, ws, wsf, plan, _lib.stream_of(x))
        else:
            L.call("omni_stem_conv_wgrad", _lib.ptr(xv), _lib.ptr(dv), _lib.ptr(dst), N, H, W, C, K, R, C, K, acc, _lib.stream_of(x))

    def launch(dst, acc):
        if not on:
            call(dst, acc, None, 0, None)
            return
        plan, addr = _det.new_plan()
        call(dst, acc, None, 0, addr)
        ws = torch.empty(max(int(plan[3]), 1), dtype=torch.float32, device=x.device)
        call(dst, acc, _lib.ptr(ws), int(plan[3]), None)
    if accum_into is not None:
        gv = accum_into.permute(0, 2, 3, 1)
        assert gv.is_contiguous() and tuple(gv.shape) == (K, R, R, C)
        launch(gv, 1)
        return None
    dw = torch.empty((K, R, R, C), dtype=torch.float32, device=x.device)
    launch(dw, 0)
    return dw.permute(0, 3, 1, 2)


# ---- depthwise convolution (csrc/depthwise.hip); weights cross as (R, R, C) tap-major ----------------------------------------------
def _taps(w):
    """(C, 1, R, R) parameter -> (R, R, C) contiguous (a few KB)"""
    C, _, R, S = w.shape
    return w.reshape(C, R, S).permute(1, 2, 0).contiguous()


def dwconv_fwd(x, w, stride=1, pad=1):
    xv, wt = _nhwc(x), _taps(w)
    N, H, W, C = xv.shape
    R = wt.shape[0]
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    L = _lib.check_device(xv, wt)
    out = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    L.call("omni_dwconv_fwd", _lib.ptr(xv), _lib.ptr(wt), _lib.ptr(out), N, H, W, C, R, stride, pad, _lib.stream_of(x))
    return out.permute(0, 3, 1, 2)


def dwconv_dgrad(dy, w, in_hw, stride=1, pad=1):
    dyv, wt = _nhwc(dy), _taps(w)
    N, _, _, C = dyv.shape
    H, W = in_hw
    L = _lib.check_device(dyv, wt)
    dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    L.call("omni_dwconv_dgrad", _lib.ptr(dyv), _lib.ptr(wt), _lib.ptr(dx), N, H, W, C, wt.shape[0], stride, pad, _lib.stream_of(dy))
    return dx.permute(0, 3, 1, 2)


def dwconv_wgrad(x, dy, R, stride=1, pad=1):
    """-> dw in the parameter's shape (C, 1, R, R)"""
    xv, dyv = _nhwc(x), _nhwc(dy)
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, dyv)
    dw = torch.empty((R, R, C), dtype=torch.float32, device=x.device)
    L.call("omni_dwconv_wgrad", _lib.ptr(xv), _lib.ptr(dyv), _lib.ptr(dw), N, H, W, C, R, stride, pad, _lib.stream_of(x))
    return dw.permute(2, 0, 1).reshape(C, 1, R, R)


# ---- grouped convolution (one implicit-GEMM launch per group on channel slices) ----------------------------------------------------
def grouped_conv2d_fwd(x, w, groups, stride=1, pad=0):
    """x (N,C,H,W) CL, w (K, C/groups, R, S) CL -> y (N,K,OH,OW) CL"""
    xv, wv = _nhwc(x), _nhwc(w)
    N, H, W, C = xv.shape
    K, R, S, Cg = wv.shape
    assert Cg * groups == C, (C, Cg, groups)
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    L = _lib.check_device(xv, wv)
    out = torch.empty((N, OH, OW, K), dtype=torch.float32, device=x.device)
    L.call("omni_grouped_conv2d_fwd", _lib.ptr(xv), _lib.ptr(wv), _lib.ptr(out), N, H, W, C, K, R, S, stride, pad, groups, _lib.stream_of(x))
    return out.permute(0, 3, 1, 2)


def grouped_conv2d_dgrad(dy, w, groups, in_hw, stride=1, pad=0):
    dyv, wv = _nhwc(dy), _nhwc(w)
    N, _, _, K = dyv.shape
    _, R, S, Cg = wv.shape
    H, W = in_hw
    L = _lib.check_device(dyv, wv)
    dx = torch.empty((N, H, W, Cg * groups), dtype=torch.float32, device=dy.device)
    L.call("omni_grouped_conv2d_dgrad", _lib.ptr(dyv), _lib.ptr(wv), _lib.ptr(dx), N, H, W, Cg * groups, K, R, S, stride, pad, groups,
           _lib.stream_of(dy))
    return dx.permute(0, 3, 1, 2)


def grouped_conv2d_wgrad(x, dy, groups, ksize, stride=1, pad=0):
    xv, dyv = _nhwc(x), _nhwc(dy)
    N, H, W, C = xv.shape
    K = dyv.shape[3]
    R, S = ksize
    L = _lib.check_device(xv, dyv)
    dw = torch.empty((K, R, S, C // groups), dtype=torch.float32, device=x.device)
    L.call("omni_grouped_conv2d_wgrad", _lib.ptr(xv), _lib.ptr(dyv), _lib.ptr(dw), N, H, W, C, K, R, S, stride, pad, groups, _lib.stream_of(x))
    return dw.permute(0, 3, 1, 2)
