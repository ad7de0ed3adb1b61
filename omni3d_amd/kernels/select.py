"""Launchers for csrc/select_nms.hip (sorted top-k per row, greedy NMS)."""
import torch

from .. import lib as _lib


def topk_rows(keys, k, n=None, pitch=None, estride=1):
    """keys: 2-D (rows, n) float32 (or a flat buffer addressed by pitch/estride).
    -> (vals (rows,k) desc, idx (rows,k) int32); ties resolve to the lower index."""
    if keys.dim() == 2 and n is None:
        rows, n = keys.shape
        pitch = keys.stride(0)
        estride = keys.stride(1) if n > 1 else 1
    else:
        rows = keys.shape[0]
    L = _lib.get()
    if not keys.is_cuda and not L.emulated:
        raise _lib.OmniHipError("omni3d_amd ops run on the GPU only")
    vals = torch.empty((rows, k), dtype=torch.float32, device=keys.device)
    idx = torch.empty((rows, k), dtype=torch.int32, device=keys.device)
    L.call("omni_topk_rows", _lib.ptr(keys), rows, n, pitch, estride, k, _lib.ptr(vals), _lib.ptr(idx),
           _lib.stream_of(keys))
    return vals, idx


def topk_segments(keys, seg_off, seg_n, k):
    """keys (rows, n) float32; -> (vals, idx) of shape (rows, nseg, k): the sorted top-k of every column segment
    [seg_off[s], seg_off[s] + seg_n[s]) of every row, ONE launch; indices are relative to the segment start."""
    import ctypes
    rows = keys.shape[0]
    L = _lib.get()
    if not keys.is_cuda and not L.emulated:
        raise _lib.OmniHipError("omni3d_amd ops run on the GPU only")
    S = len(seg_off)
    vals = torch.empty((rows, S, k), dtype=torch.float32, device=keys.device)
    idx = torch.empty((rows, S, k), dtype=torch.int32, device=keys.device)
    offs = (ctypes.c_int * S)(*[int(v) for v in seg_off])
    ns = (ctypes.c_int * S)(*[int(v) for v in seg_n])
    L.call("omni_topk_segments", _lib.ptr(keys), rows, keys.stride(0), keys.stride(1) if keys.shape[1] > 1 else 1, S,
           ctypes.cast(offs, ctypes.c_void_p), ctypes.cast(ns, ctypes.c_void_p), k, _lib.ptr(vals), _lib.ptr(idx),
           _lib.stream_of(keys))
    return vals, idx


def nms_sorted(boxes, iou_thr, counts=None, valid=None, _poison=False):
    """boxes (Q, nmax, 4) sorted by descending score -> keep (Q, nmax) int32.  (_poison: tests fill the scratch and the output
    with garbage first -- the kernels must not depend on their previous contents.)"""
    boxes = boxes.contiguous()
    Q, nmax, _ = boxes.shape
    L = _lib.check_device(boxes, counts, valid)
    words = (nmax + 63) // 64
    ws = torch.empty(max(Q * nmax * words, 1), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((Q, nmax), dtype=torch.int32, device=boxes.device)
    if _poison:
        ws.fill_(-1)
        keep.fill_(7)
    L.call("omni_nms_sorted", _lib.ptr(boxes), _lib.ptr(counts), _lib.ptr(valid), Q, nmax, float(iou_thr), _lib.ptr(ws),
           _lib.ptr(keep), _lib.stream_of(boxes))
    return keep
