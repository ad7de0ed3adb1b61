"""csrc/gemm_engine.hip (LDS-DMA ring, cross-barrier fragment prefetch, persistent workgroups) vs torch matmul in float64:
every form (NT / NN / TN), both tiles, ragged M / N, K tails, batched, split-K, accumulate, bias + ReLU, and several work items
per persistent workgroup (the ring runs across item boundaries)."""
import pytest
import torch

CASES = [  # form, batch, M, N, K, tile, splits, workgroups
    ("NT", 1, 130, 72, 96, 2, 1, 0),        # ragged M and N tiles
    ("NT", 3, 200, 136, 64, 2, 1, 8),       # batched, 8 persistent workgroups => several items each, mixed batches
    ("NT", 1, 300, 130, 40, 1, 1, 0),       # 256x128 tile, K tail (40 = 32 + 8)
    ("NT", 1, 64, 64, 416, 2, 3, 0),        # split-K (13 slabs over 3 splits), atomic epilogue
    ("NN", 1, 150, 96, 72, 2, 1, 0),        # data-gradient form, K tail
    ("NN", 2, 260, 132, 64, 1, 1, 8),
    ("TN", 1, 72, 136, 200, 2, 1, 0),       # weight-gradient form, reduction over 200 "pixels" (tail slab)
    ("TN", 2, 264, 68, 96, 1, 2, 8),
    # balanced (-1): whole tiles + one part of the left-over tiles per workgroup
    ("TN", 1, 520, 392, 200, 2, -1, 8),     # 5 x 4 = 20 tiles on 8 workgroups: 2 whole tiles each, 4 tiles cut in 2 (7 slabs -> 4 + 3)
    ("NT", 2, 300, 200, 136, 2, -1, 16),    # 2 x 3 x 2 = 12 tiles on 16 workgroups: no whole tile, 12 parts (ts = 1), batched
    ("NT", 1, 130, 72, 416, 2, -1, 8),      # 2 tiles on 8 workgroups: both cut in 4 (13 slabs -> 4 + 4 + 4 + 1)
    ("NN", 1, 700, 260, 72, 1, -1, 8),      # 256x128 tiles: 3 x 3 = 9 on 8: one whole tile each, 1 tile cut in 3 (3 slabs)
    ("NT", 1, 256, 256, 64, 2, -1, 8),      # 4 tiles on 8 workgroups, 2 slabs: cut in 2
    ("TN", 1, 256, 512, 64, 2, -1, 8),      # 8 tiles on 8 workgroups: exact rounds, plain launch
]


def _run(dev):
    from omni3d_amd.kernels import gemm as G
    g = torch.Generator().manual_seed(0)
    for form, b, M, N, K, tile, splits, wgs in CASES:
        if form == "NT":
            A, B = torch.randn(b, M, K, generator=g), torch.randn(b, N, K, generator=g)
            ref = torch.einsum("bmk,bnk->bmn", A.double(), B.double())
        elif form == "NN":
            A, B = torch.randn(b, M, K, generator=g), torch.randn(b, K, N, generator=g)
            ref = torch.einsum("bmk,bkn->bmn", A.double(), B.double())
        else:
            A, B = torch.randn(b, K, M, generator=g), torch.randn(b, K, N, generator=g)
            ref = torch.einsum("bkm,bkn->bmn", A.double(), B.double())
        f = getattr(G, form)
        out = G.gemm(A.to(dev), B.to(dev), f, splits=splits, tile=tile, workgroups=wgs)
        err = float((out.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * float(ref.abs().max()) + 1e-6, (form, b, M, N, K, tile, splits, wgs, err)
    # bias + ReLU epilogue, and accumulation into an existing tensor
    A, B, bias = torch.randn(90, 64, generator=g), torch.randn(70, 64, generator=g), torch.randn(70, generator=g)
    out = G.gemm(A.to(dev), B.to(dev), G.NT, bias=bias.to(dev), relu=True, tile=2)
    ref = (A.double() @ B.double().t() + bias.double()).clamp(min=0)
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-4
    base = torch.randn(90, 70, generator=g)
    acc = base.clone().to(dev)
    G.gemm(A.to(dev), B.to(dev), G.NT, out=acc, accumulate=True, tile=2)
    assert float((acc.cpu().double() - (base.double() + A.double() @ B.double().t())).abs().max()) <= 1e-4


def test_gemm_engine_emulated(emu_lib):
    _run("cpu")


@pytest.mark.gpu
def test_gemm_engine_gpu(hip_lib):
    _run("cuda")


def test_gemm_engine_rejects_bad_arguments(emu_lib):
    from omni3d_amd.kernels import gemm as G
    from omni3d_amd.lib import OmniHipError
    with pytest.raises(OmniHipError):
        G.gemm(torch.zeros(8, 6), torch.zeros(8, 6), G.NT)          # K not a multiple of 4
    with pytest.raises(OmniHipError):
        G.gemm(torch.zeros(8, 8), torch.zeros(8, 8), G.NT, tile=7)
