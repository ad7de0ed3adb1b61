"""Deterministic mode of the split reductions (csrc/split_reduce.h): workspace + arrival-counter plumbing.

The reference's PyTorch-CPU path is run-to-run identical; rounds 1-3 of this path were not (fp32 atomics where a reduction is cut
over several workgroups).  With the switch on, every launcher that would split a reduction asks the C launcher for its plan
(`plan` argument of the `*_det` entry points: the launcher's own tile / split decision, no duplicated heuristics here), takes a
workspace for the partial tiles from torch's caching allocator (stream-ordered; inside a hipGraph capture from that graph's pool,
so the critical-path graphs and the weight-gradient graphs that replay beside them never share one) and hands over a block of
arrival counters that the kernels leave zeroed.

Counters are persistent per *domain*: launches that can run concurrently must not share them.  Eager launches are keyed by the
stream they go to; captured launches by the graph family they belong to (`domain("W")` around the capture of a weight-gradient
graph, "M" otherwise) -- graphs of one family replay on one stream, one after another.

OMNI_DETERMINISTIC=0 returns to the atomic epilogues (A/B switch; the `-m gpu` tests and bench.py run with the default)."""
import ctypes
import os

import torch

_ON = os.environ.get("OMNI_DETERMINISTIC", "1") != "0"
_CTR_CAP = 1 << 16
_domain = None
_counters = {}


def on():
    return _ON


def set_enabled(flag):
    """-> previous setting (tests / A-B tools)"""
    global _ON
    prev, _ON = _ON, bool(flag)
    return prev


class domain:
    """with detmode.domain("W"): ...   -- launches captured inside belong to that graph family"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _domain
        self.prev, _domain = _domain, self.name

    def __exit__(self, *a):
        global _domain
        _domain = self.prev


def _key(like):
    if like.is_cuda:
        if torch.cuda.is_current_stream_capturing():
            return (like.device.index, "graph", _domain or "M")
        if _domain is not None:
            return (like.device.index, "domain", _domain)
        return (like.device.index, "stream", torch.cuda.current_stream(like.device).cuda_stream)
    return ("cpu", _domain)


def counters(like, n):
    """-> int32 tensor of >= n zeroed arrival counters for a launch on `like`'s device and the current stream / graph family"""
    if n > _CTR_CAP:
        return torch.zeros(n, dtype=torch.int32, device=like.device)
    key = _key(like)
    c = _counters.get(key)
    if c is None:
        if like.is_cuda and torch.cuda.is_current_stream_capturing():
            # first use of this family inside a capture: the buffer must outlive the graph's private pool -> allocate it outside
            # the capture's allocator scope is not possible here, so it becomes part of the capture (zeroed by a captured fill on
            # every replay, which is harmless: the kernels leave it zeroed anyway)
            c = torch.zeros(_CTR_CAP, dtype=torch.int32, device=like.device)
            return c
        c = torch.zeros(_CTR_CAP, dtype=torch.int32, device=like.device)
        _counters[key] = c
    return c


def nonzero_counters():
    """-> [(key, non-zero entries)] over every persistent arrival-counter block (host sync: tests and OMNI_DET_SELFCHECK only).  The
    deterministic split reductions rely on every counter being zero on entry and leave it zero on exit (csrc/split_reduce.h); a launch
    that faulted, or whose plan and launch calls disagreed on the split count, would leave one behind and silently corrupt every
    later reduction of that stream / graph family.  tests/test_determinism.py asserts this list is empty after full
    training steps; OMNI_DET_SELFCHECK=1 checks the block a launch is about to use and raises."""
    return [(k, int((c != 0).sum())) for k, c in _counters.items() if bool((c != 0).any())]


_SELFCHECK = os.environ.get("OMNI_DET_SELFCHECK", "0") == "1"


def prewarm(device, families=("M", "W")):
    """allocate the graph families' counter blocks BEFORE a capture starts (graphed.py)"""
    dev = torch.device(device)
    if dev.type != "cuda":
        return
    for f in families:
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), "graph", f)
        if key not in _counters:
            _counters[key] = torch.zeros(_CTR_CAP, dtype=torch.int32, device=dev)


def new_plan():
    buf = (ctypes.c_longlong * 4)()
    return buf, ctypes.addressof(buf)


def workspace(like, plan):
    """plan = the four values a `*_det` entry point reported -> (ws tensor or None, ws_floats, counter tensor, n_ctr)"""
    n_ctr, ws_floats = int(plan[2]), int(plan[3])
    ws = torch.empty(ws_floats, dtype=torch.float32, device=like.device) if ws_floats > 0 else None
    if n_ctr <= 0 and ws_floats <= 0:
        # no split: the launch needs neither slots nor counters, only a non-null `ctr` to select the deterministic entry's behaviour --
        # any already-existing block will do, and nothing is allocated (inside a capture: no extra fill node)
        existing = _counters.get(_key(like))
        if existing is not None:
            return None, 0, existing, 1
    ctr = counters(like, max(n_ctr, 1))
    if _SELFCHECK and like.is_cuda and not torch.cuda.is_current_stream_capturing() and bool((ctr != 0).any()):
        raise RuntimeError("omni3d_amd: arrival counters of a deterministic reduction are not zero on entry (an earlier launch of this "
                           f"stream faulted or was planned with another split count): {int((ctr != 0).sum())} entries")
    return ws, ws_floats, ctr, max(n_ctr, 1)
