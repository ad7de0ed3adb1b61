"""Top-k / NMS kernels (csrc/select_nms.hip): selections must be index-exact vs the oracle
(oracle/upstream.py nms, torch stable sort), including adversarial ties."""
import numpy as np
import pytest
import torch

from oracle import upstream as U


def _topk_case(dev, rows, n, k, seed, ties=False):
    from omni3d_amd.kernels import select
    g = torch.Generator().manual_seed(seed)
    keys = torch.randn(rows, n, generator=g)
    if ties:
        keys = (keys * 2).round() / 2          # heavy duplication
        keys[0, : n // 2] = float("-inf")      # masked entries
    vals, idx = select.topk_rows(keys.to(dev), k)
    sv, si = torch.sort(keys, dim=1, descending=True, stable=True)
    kk = min(k, n)
    assert torch.equal(idx.cpu()[:, :kk].long(), si[:, :kk])
    assert torch.equal(vals.cpu()[:, :kk], sv[:, :kk])
    if k > n:
        assert (idx.cpu()[:, n:] == -1).all()


def _topk_prefilter_overflow(dev):
    """the two-level selection's bound comes from the per-thread top 4 (thread = index mod 1024); when 64 threads own ALL the large
    values the bound is far too low, more than 4096 elements pass it and the segment must take the full selection instead"""
    from omni3d_amd.kernels import select
    g = torch.Generator().manual_seed(9)
    n, k = 65536, 3000
    keys = torch.rand(2, n, generator=g)
    hot = (torch.arange(n) % 1024) < 64
    keys[:, hot] += 10.0
    keys[1, ::3] = keys[1, 5]                      # and ties
    vals, idx = select.topk_rows(keys.to(dev), k)
    sv, si = torch.sort(keys, dim=1, descending=True, stable=True)
    assert torch.equal(idx.cpu().long(), si[:, :k]) and torch.equal(vals.cpu(), sv[:, :k])


def _strided_topk(dev):
    from omni3d_amd.kernels import select
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(3, 100, 16, generator=g)   # logits interleaved with other channels, pitch 16
    flat = buf.to(dev)
    for a in range(3):
        view = flat[:, :, a]
        vals, idx = select.topk_rows(view, 7)
        sv, si = torch.sort(buf[:, :, a], dim=1, descending=True, stable=True)
        assert torch.equal(idx.cpu().long(), si[:, :7])


def _rand_boxes(g, n, size=100.0):
    xy = torch.rand(n, 2, generator=g) * size
    wh = torch.rand(n, 2, generator=g) * size * 0.4 + 1.0
    return torch.cat([xy, xy + wh], dim=1)


def _nms_case(dev, Q, nmax, thr, seed):
    from omni3d_amd.kernels import select
    g = torch.Generator().manual_seed(seed)
    boxes = torch.stack([_rand_boxes(g, nmax) for _ in range(Q)])
    boxes[0, 3] = boxes[0, 2]                       # duplicate box: IoU == 1 > thr
    counts = torch.randint(max(nmax // 2, 1), nmax + 1, (Q,), generator=g).int()
    counts[0] = nmax
    valid = (torch.rand(Q, nmax, generator=g) > 0.1).int()
    keep = select.nms_sorted(boxes.to(dev), thr, counts.to(dev), valid.to(dev), _poison=True).cpu()
    for q in range(Q):
        n = int(counts[q])
        sel = torch.where(valid[q, :n] != 0)[0]
        scores = torch.arange(n, 0, -1, dtype=torch.float32)[sel]   # already sorted by score
        ref = sel[U.nms(boxes[q, :n][sel], scores, thr)]
        got = torch.where(keep[q, :n] != 0)[0]
        assert torch.equal(got, torch.sort(ref)[0]), (q, got, ref)
        assert (keep[q, n:] == 0).all()


def _segments_case(dev, rows, seg_n, k, seed):
    """one launch over column segments == topk_rows on each slice (the per-FPN-level pre-NMS top-k)"""
    from omni3d_amd.kernels import select
    g = torch.Generator().manual_seed(seed)
    n = sum(seg_n)
    keys = torch.randn(rows, n, generator=g)
    keys[:, ::7] = keys[:, 3:4]                       # ties across the row
    keys = keys.to(dev)
    off = [sum(seg_n[:i]) for i in range(len(seg_n))]
    v, i = select.topk_segments(keys, off, seg_n, k)
    assert v.shape == (rows, len(seg_n), k)
    for s, (o, m) in enumerate(zip(off, seg_n)):
        rv, ri = select.topk_rows(keys[:, o:o + m], k)
        assert torch.equal(v[:, s], rv) and torch.equal(i[:, s], ri), s


def test_topk_segments_emulated(emu_lib):
    _segments_case("cpu", 2, [700, 190, 48, 12, 3], 64, 0)


@pytest.mark.gpu
def test_topk_segments_gpu(hip_lib):
    _segments_case("cuda", 4, [49152, 12288, 3072, 768, 192], 2000, 1)


def test_topk_emulated(emu_lib):
    _topk_case("cpu", 2, 5000, 300, 0)
    _topk_case("cpu", 2, 700, 2000, 1)            # k > n
    _topk_case("cpu", 3, 3000, 64, 2, ties=True)
    _topk_case("cpu", 1, 66000, 100, 3, ties=True)    # > 65536 elements: the streamed variant (rows are not cached in registers)
    _topk_case("cpu", 1, 65536, 40, 4)                # the largest cached row
    _topk_case("cpu", 1, 9000, 3072, 5, ties=True)    # the largest k of the two-level selection
    _topk_case("cpu", 1, 9000, 3073, 6, ties=True)    # one more: cached full selection
    _topk_prefilter_overflow("cpu")
    _strided_topk("cpu")


def test_nms_emulated(emu_lib):
    _nms_case("cpu", 2, 150, 0.5, 0)
    _nms_case("cpu", 1, 64, 0.7, 1)
    _nms_case("cpu", 2, 700, 0.5, 2)              # 11 chunks: the per-wave removed sets across many chunks
    _nms_case("cpu", 1, 4300, 0.6, 3)             # more than 64 words: the second register word of the removed set


@pytest.mark.gpu
def test_topk_gpu(hip_lib):
    _topk_case("cuda", 20, 49152, 2000, 0)
    _topk_case("cuda", 4, 65472, 256, 1, ties=True)
    _topk_case("cuda", 4, 192, 2000, 2)
    _topk_case("cuda", 4, 6960, 1000, 3, ties=True)
    _topk_case("cuda", 2, 70000, 300, 4, ties=True)   # streamed variant
    _topk_case("cuda", 2, 65536, 8192, 5)
    _topk_case("cuda", 3, 30000, 3072, 6, ties=True)
    _topk_prefilter_overflow("cuda")
    _strided_topk("cuda")


@pytest.mark.gpu
def test_nms_gpu(hip_lib):
    _nms_case("cuda", 20, 2000, 0.7, 0)
    _nms_case("cuda", 3, 777, 0.5, 1)
    _nms_case("cuda", 2, 8192, 0.6, 2)            # the largest problem: 128 words
