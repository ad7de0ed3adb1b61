"""Launcher of csrc/gemm_engine.hip (the round-2 fp32-MFMA GEMM main loop): NT / NN / TN, batched, split-K, persistent."""
import torch

from .. import lib as _lib

import os

NT, NN, TN = 0, 1, 2
BALANCED = -1
# Multi-GPU readiness: the engine's launches are PERSISTENT -- one workgroup per CU for the whole GEMM (0.34-0.47 ms
# for the fc1-class ones) -- and a collective's kernel cannot start on a CU whose LDS / wave slots such a workgroup holds.  In the
# staged step every engine launch sits in stage 0, before the first exchange is issued (solver/graphed.py: stage k's ranges leave
# behind W_k), so nothing should collide; if an 8-GPU profile shows RCCL kernels queueing behind them anyway, OMNI_ENGINE_WGS=240
# leaves 16 CUs free (a multiple of 8: the balanced split needs one).  It is NOT the default: the balanced split cuts its left-over
# tiles by the workgroup count, so another count is another fp32 summation order, and no run over RCCL has shown the need.
ENGINE_WGS = int(os.environ.get("OMNI_ENGINE_WGS", "0"))


def gemm(A, B, form=NT, bias=None, relu=False, out=None, accumulate=False, splits=1, tile=1, workgroups=0):
    """A, B: 2-D or 3-D (batched) contiguous fp32 tensors.
       NT: A (b, M, K), B (b, N, K) -> (b, M, N)      NN: A (b, M, K), B (b, K, N)      TN: A (b, K, M), B (b, K, N) -> (b, M, N)
    out: optional preallocated result (accumulate=True adds into it with fp32 atomics); splits > 1 zeroes `out` first unless
    accumulating.  splits = BALANCED (-1): whole tiles per workgroup plus one part of the left-over tiles each (csrc/gemm_engine.hip);
    without the deterministic mode `out` is zeroed when tiles are cut (their parts meet through atomics)."""
    workgroups = workgroups or ENGINE_WGS
    squeeze = A.dim() == 2
    A3, B3 = (A.unsqueeze(0), B.unsqueeze(0)) if squeeze else (A, B)
    assert A3.is_contiguous() and B3.is_contiguous() and A3.shape[0] == B3.shape[0]
    b = A3.shape[0]
    if form == NT:
        M, K = A3.shape[1:]
        N = B3.shape[1]
        assert B3.shape[2] == K
        lda, ldb = K, K
    elif form == NN:
        M, K = A3.shape[1:]
        N = B3.shape[2]
        assert B3.shape[1] == K
        lda, ldb = K, N
    else:
        K, M = A3.shape[1:]
        N = B3.shape[2]
        assert B3.shape[1] == K
        lda, ldb = M, N
    L = _lib.check_device(A3, B3, bias, out)
    if out is None:
        out3 = torch.empty((b, M, N), dtype=torch.float32, device=A.device)
    else:
        out3 = out.unsqueeze(0) if out.dim() == 2 else out
        assert out3.is_contiguous() and tuple(out3.shape) == (b, M, N)
    from . import detmode as _det
    if _det.on():
        args = (_lib.ptr(A3), _lib.ptr(B3), _lib.ptr(out3), _lib.ptr(bias), form, b, M, N, K, lda, ldb, N, A3.stride(0), B3.stride(0), M * N,
                splits, int(relu), int(accumulate), tile, workgroups)
        plan, addr = _det.new_plan()
        L.call("omni_gemm_engine_det", *args, None, 0, None, 0, addr, _lib.stream_of(A))
        ws, wsf, ctr, nctr = _det.workspace(A, plan)
        L.call("omni_gemm_engine_det", *args, _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, _lib.stream_of(A))
        return out3[0] if squeeze else out3
    if splits > 1 and not accumulate:
        out3.zero_()
    elif splits == BALANCED and not accumulate and tile == 2:      # (tile 1 has no balanced form: the launcher runs whole tiles)
        W = workgroups or 256
        bm, bn = (128, 128)
        tm, tn = (M + bm - 1) // bm, (N + bn - 1) // bn
        if (b * tm * tn) % W:
            out3.zero_()        # (the cut tiles are the last ones of the engine's banded tile order: not a suffix of rows)
    L.call("omni_gemm_engine", _lib.ptr(A3), _lib.ptr(B3), _lib.ptr(out3), _lib.ptr(bias), form, b, M, N, K, lda, ldb, N,
           A3.stride(0), B3.stride(0), M * N, splits, int(relu), int(accumulate), tile, workgroups, _lib.stream_of(A))
    return out3[0] if squeeze else out3


def transpose2d(src):
    """(rows, cols) contiguous -> (cols, rows) contiguous"""
    assert src.dim() == 2 and src.is_contiguous() and src.dtype == torch.float32
    L = _lib.check_device(src)
    dst = torch.empty((src.shape[1], src.shape[0]), dtype=torch.float32, device=src.device)
    L.call("omni_transpose2d", _lib.ptr(src), _lib.ptr(dst), src.shape[0], src.shape[1], _lib.stream_of(src))
    return dst
