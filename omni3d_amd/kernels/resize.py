"""Launcher of csrc/resize.hip: PIL-exact bilinear resampling (+ horizontal flip) of uint8 images on the device.

The coefficient rows are Pillow's (`precompute_coeffs` + `normalize_coeffs_8bpc`, Resample.c): triangle filter, support
scaled by the down-scale factor, normalised in double precision, converted to 22-bit fixed point with round-half-away."""
import math
from functools import lru_cache

import numpy as np
import torch

from .. import lib as _lib

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=256)
def pil_bilinear_coeffs(in_size, out_size):
    """-> (bounds (out,2) int32 [first, count], kk (out, ksize) int32, ksize) for resampling in_size -> out_size"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                    # bilinear: filter support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0 else t
            v = 1.0 - t if t < 1.0 else 0.0
            w[x] = v
            ww += v
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(v - 0.5) if w[x] < 0 else int(v + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_bilinear_u8(img, out_h, out_w, flip=False):
    """img (..., H, W) uint8 tensor (any leading plane dims) -> (..., out_h, out_w) uint8, == PIL Image.resize(BILINEAR)
    followed by an optional horizontal mirror (detectron2 ResizeTransform + HFlipTransform)."""
    assert img.dtype == torch.uint8
    src = img.contiguous()
    H, W = src.shape[-2:]
    planes = src.numel() // (H * W)
    L = _lib.check_device(src)
    dev = src.device
    bh, kh, ksh = pil_bilinear_coeffs(W, out_w)
    bv, kv, ksv = pil_bilinear_coeffs(H, out_h)
    tb = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731  (a few KB per distinct size pair)
    bh_t, kh_t, bv_t, kv_t = tb(bh), tb(kh), tb(bv), tb(kv)
    out = torch.empty(src.shape[:-2] + (out_h, out_w), dtype=torch.uint8, device=dev)
    tmp = torch.empty(planes * H * out_w, dtype=torch.uint8, device=dev)
    L.call("omni_resize_bilinear_u8", _lib.ptr(src), _lib.ptr(out), _lib.ptr(tmp), planes, H, W, out_h, out_w, _lib.ptr(bh_t),
           _lib.ptr(kh_t), ksh, _lib.ptr(bv_t), _lib.ptr(kv_t), ksv, int(flip), _lib.stream_of(src))
    return out
