"""Launchers of csrc/glue.hip (round 6): the scalar-sized launches around the losses and the loop, one kernel each -- and the
device-side state of the in-kernel subsampling draws (csrc/philox.h)."""
import ctypes

import torch

from .. import lib as _lib

_P = ctypes.c_void_p
MAXV = 16


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def scale_vec(src, coef=None, denom=None, denom_min=1.0, n=None, out=None):
    """out[i] = float(src[i] * coef[i] / max(denom, denom_min)) for the first n elements of `src` (float32 or float64, 1-D; a stride-0
    view = one broadcast scalar); coef: n Python floats or None; denom: one-element tensor of src's dtype or None.  One launch."""
    L = _lib.check_device(src.contiguous() if src.stride(0) not in (0, 1) else src.detach()[:1])
    n = int(n if n is not None else src.shape[0])
    assert 0 < n <= MAXV and src.dim() == 1 and src.dtype in (torch.float32, torch.float64)
    stride = int(src.stride(0))
    assert stride in (0, 1)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=src.device)
    c = (ctypes.c_double * n)(*[float(v) for v in coef]) if coef is not None else None
    if denom is not None:
        assert denom.dtype == src.dtype and denom.numel() == 1
    L.call("omni_scale_vec", src.data_ptr(), int(src.dtype == torch.float64), stride, ctypes.cast(c, _P) if c is not None else None,
           denom.data_ptr() if denom is not None else None, float(denom_min), n, out.data_ptr(), _lib.stream_of(src))
    return out


def sum_vectors(first, rest=()):
    """-> (3,) float32 [sum of the elements of the vectors in `first`, of those in `rest`, of all]; one launch, fixed order"""
    vecs = [v.detach() for v in list(first) + list(rest)]
    assert 0 < len(vecs) <= MAXV
    for v in vecs:
        assert v.dtype == torch.float32 and v.is_contiguous()
    L = _lib.check_device(*vecs)
    out = torch.empty(3, dtype=torch.float32, device=vecs[0].device)
    p = _ptr_array(vecs)
    lens = (ctypes.c_int * len(vecs))(*[int(v.numel()) for v in vecs])
    L.call("omni_sum_vectors", ctypes.cast(p, _P), ctypes.cast(lens, _P), len(vecs), len(list(first)), out.data_ptr(), _lib.stream_of(vecs[0]))
    return out


def guard_gather(scalars, vec):
    """vec[i] = scalars[i] (0-d / one-element float32 device tensors), vec[n] = their sum.  -> False if the inputs do not qualify
    (the caller then stacks them the old way)"""
    n = len(scalars)
    if not (0 < n <= MAXV) or any(s.dtype != torch.float32 or s.numel() != 1 or s.device != vec.device for s in scalars):
        return False
    L = _lib.check_device(vec)
    p = _ptr_array([s.detach() for s in scalars])
    L.call("omni_guard_gather", ctypes.cast(p, _P), n, vec.data_ptr(), _lib.stream_of(vec))
    return True


def bump_counters(counters, delta=1):
    """`c += delta` for a list of one-element int64 device tensors; one launch per 64"""
    if not counters:
        return
    for c in counters:
        assert c.dtype == torch.int64 and c.numel() == 1
    L = _lib.check_device(*counters)
    p = _ptr_array(counters)
    L.call("omni_bump_counters", ctypes.cast(p, _P), len(counters), int(delta), _lib.stream_of(counters[0]))


def zero_(t):
    """t.zero_() as one kernel node of this library (contiguous tensors)"""
    L = _lib.check_device(t)
    L.call("omni_zero", t.data_ptr(), int(t.numel() * t.element_size()), _lib.stream_of(t))
    return t


class DrawState:
    """Device-side state of the in-kernel Exp(1) draws of one call site (csrc/philox.h): state = [seed, draw counter] (int64), ticket
    (int32, zero between launches).  The seed comes from torch's generator at first use, so `torch.manual_seed` makes the subsampling
    of a run reproducible like the reference's; every launch that draws advances the counter on the device (also under graph replay)."""

    def __init__(self):
        self.state, self.ticket = None, None

    def tensors(self, device):
        if self.state is None or self.state.device != torch.device(device):
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())      # host generator: follows torch.manual_seed
            try:        # ranks of a data-parallel job that seeded alike (bench.py, the reference's `seed_all_rng(seed + rank)` aside) must
                import torch.distributed as dist          # not subsample their shards with the SAME variates
                if dist.is_available() and dist.is_initialized():
                    seed = (seed ^ ((dist.get_rank() + 1) * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1)
            except Exception:  # noqa: BLE001
                pass
            self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
            self.ticket = torch.zeros(1, dtype=torch.int32, device=device)
        return self.state, self.ticket
