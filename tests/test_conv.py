"""Implicit-GEMM conv / linear kernels (csrc/conv_gemm.hip) vs plain PyTorch fp32 on CPU.
fp32 MFMA is an fmaf chain; the CPU reference sums in a different order, so the tolerance is
relative to sum|a*b| (1e-5 of it), far inside the 1e-4 north-star bar."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

CASES = [
    # N, H, W, C, K, R, stride, pad
    (1, 8, 8, 16, 32, 3, 1, 1),     # 256x32 tile path (K<=32)
    (2, 9, 7, 8, 48, 3, 2, 1),      # ragged M, 128x64 path, stride 2
    (1, 6, 6, 32, 72, 1, 1, 0),     # 1x1, 128x128 path with ragged N
    (1, 10, 10, 4, 16, 7, 1, 3),    # 7x7 stem shape (C padded to 4), Kd=196 (tail slab)
    (1, 6, 6, 64, 64, 3, 1, 1),     # few tiles + deep reduction: split-K path with atomic epilogue (+bias, +ReLU)
    (1, 7, 9, 32, 64, 3, 2, 1),     # stride-2 dgrad parity classes, odd extents, tap-inner (K % 32 == 0) + split-K
    (1, 7, 6, 16, 24, 1, 2, 0),     # 1x1/s2 (ResNet downsample): three of the four dgrad classes have no taps -> zeros
    (1, 12, 10, 4, 16, 7, 2, 3),    # 7x7/s2 ResNet stem
    (2, 5, 6, 16, 72, 3, 1, 1),     # wgrad 128x128 tile (K > 64, R*S*C > 64), pixel cursor wrapping rows and images
    (2, 32, 40, 64, 128, 3, 1, 1),  # wgrad: 9 tiles x 10 pixel splits (with OMNI_WGRAD_XCD_SPLITS=1: 8 of them dealt out per XCD, 2 in the plain order)
]


def _mk(dev, *shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dev)


def _check(a, b, scale):
    err = (a.cpu() - b).abs().max().item()
    assert err <= 2e-5 * scale + 1e-6, (err, scale)


def _run_case(dev, case, tile=0, splits=0):
    from omni3d_amd.kernels import conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk(dev, N, C, H, W, seed=1).contiguous(memory_format=torch.channels_last)
    w = (_mk(dev, K, C, R, R, seed=2) * 0.2).contiguous(memory_format=torch.channels_last)
    b = _mk(dev, K, seed=3)
    xc, wc, bc = x.cpu(), w.cpu(), b.cpu()
    ref = F.conv2d(xc, wc, bc, stride=stride, padding=pad)
    scale = float(F.conv2d(xc.abs(), wc.abs(), None, stride=stride, padding=pad).max())
    y = conv.conv2d_fwd(x, w, b, stride, pad, relu=False, tile=tile, splits=splits)
    assert y.shape == ref.shape
    _check(y, ref, scale)
    yr = conv.conv2d_fwd(x, w, b, stride, pad, relu=True, tile=tile, splits=splits)
    _check(yr, ref.clamp(min=0), scale)
    dy = _mk(dev, *ref.shape, seed=4).contiguous(memory_format=torch.channels_last)
    dyc = dy.cpu()
    dx_ref = torch.nn.grad.conv2d_input(xc.shape, wc, dyc, stride=stride, padding=pad)
    dw_ref = torch.nn.grad.conv2d_weight(xc, wc.shape, dyc, stride=stride, padding=pad)
    dx = conv.conv2d_dgrad(dy, w, (H, W), stride, pad, tile=tile, splits=splits)
    _check(dx, dx_ref, float(dx_ref.abs().max()) * 4 + 1)
    dw = conv.conv2d_wgrad(x, dy, (R, R), stride, pad, tile=tile)
    _check(dw, dw_ref, float(dw_ref.abs().max()) * 4 + 1)


def _run_linear(dev, M, C, K):
    from omni3d_amd.kernels import conv
    x, w, b = _mk(dev, M, C, seed=5), _mk(dev, K, C, seed=6) * 0.1, _mk(dev, K, seed=7)
    ref = F.linear(x.cpu(), w.cpu(), b.cpu())
    y = conv.linear_fwd(x, w, b, relu=True)
    _check(y, ref.clamp(min=0), float(ref.abs().max()) * 4)
    dy = _mk(dev, M, K, seed=8)
    _check(conv.linear_dgrad(dy, w), dy.cpu() @ w.cpu(), float((dy.cpu() @ w.cpu()).abs().max()) * 4)
    _check(conv.linear_wgrad(x, dy), dy.cpu().t() @ x.cpu(), float((dy.cpu().t() @ x.cpu()).abs().max()) * 4)


@pytest.mark.parametrize("case", CASES)
def test_conv_emulated(emu_lib, case):
    _run_case("cpu", case)


@pytest.mark.parametrize("tile,splits", [(1, 1), (1, 2), (2, 3), (3, 1), (4, 1)])
def test_conv_explicit_algo_emulated(emu_lib, tile, splits):
    """every tile shape of the fwd / dgrad / wgrad kernels, with and without a split reduction, requested explicitly through
    the *_algo entry points on a small shape (the automatic choice would never pick 128x128 here)"""
    _run_case("cpu", (1, 12, 12, 16, 72, 3, 1, 1), tile=tile, splits=splits)


def test_conv_algo_rejects_bad_arguments(emu_lib):
    from omni3d_amd.kernels import conv
    from omni3d_amd.lib import OmniHipError
    x = torch.zeros(1, 16, 8, 8).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(8, 16, 3, 3).contiguous(memory_format=torch.channels_last)
    with pytest.raises(OmniHipError):
        conv.conv2d_fwd(x, w, None, 1, 1, tile=9)


def _run_transposed_dgrad(dev, M, K, C):
    """the fc1-class data gradient as dY (W^T)^T: the LDS-tiled transpose (ragged 64x64 tiles) + the engine's NT form, the way
    conv.linear_dgrad runs it for M >= 512, C >= 4096, K >= 512"""
    from omni3d_amd.kernels import gemm as G
    g = torch.Generator().manual_seed(12)
    w = (torch.randn(K, C, generator=g) * 0.1).to(dev)
    dy = torch.randn(M, K, generator=g).to(dev)
    wt = G.transpose2d(w)
    assert wt.shape == (C, K) and torch.equal(wt.cpu(), w.cpu().t().contiguous())
    ref = dy.cpu().double() @ w.cpu().double()
    for splits in (1, G.BALANCED):             # plain launch; whole tiles + parts of the left-over tiles (cut tiles pre-zeroed)
        dx = G.gemm(dy, wt, G.NT, tile=2, splits=splits, workgroups=8 if dev == "cpu" else 0)
        assert (dx.cpu().double() - ref).abs().max() <= 2e-6 * float((dy.cpu().abs() @ w.cpu().abs()).max()), splits


def _run_fc1_class_gradients(dev, M=1024, C=4096, K=512):
    """shapes that take the engine routes of conv.linear_dgrad / conv.linear_wgrad (accumulating form of the training step)"""
    from omni3d_amd.kernels import conv
    g = torch.Generator().manual_seed(3)
    x, w, dy = torch.randn(M, C, generator=g).to(dev), (torch.randn(K, C, generator=g) * 0.05).to(dev), torch.randn(M, K, generator=g).to(dev)
    base = torch.randn(K, C, generator=g)
    acc = base.clone().to(dev)
    assert conv.linear_wgrad(x, dy, accum_into=acc) is None
    ref = base.double() + dy.cpu().double().t() @ x.cpu().double()
    assert (acc.cpu().double() - ref).abs().max() <= 2e-6 * float((dy.cpu().abs().t() @ x.cpu().abs()).max())
    dx = conv.linear_dgrad(dy, w)
    ref = dy.cpu().double() @ w.cpu().double()
    assert (dx.cpu().double() - ref).abs().max() <= 2e-6 * float((dy.cpu().abs() @ w.cpu().abs()).max())


def test_linear_emulated(emu_lib):
    _run_linear("cpu", 70, 36, 20)
    _run_transposed_dgrad("cpu", 70, 96, 200)
    _run_transposed_dgrad("cpu", 300, 64, 520)      # 3 x 5 = 15 tiles on 8 workgroups: one whole tile each, 7 left over


def test_conv_rejects_unaligned_channels(emu_lib):
    from omni3d_amd.kernels import conv
    from omni3d_amd.lib import OmniHipError
    x = torch.zeros(1, 3, 8, 8).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(8, 3, 3, 3).contiguous(memory_format=torch.channels_last)
    with pytest.raises(OmniHipError):
        conv.conv2d_fwd(x, w, None, 1, 1)


GPU_CASES = CASES + [
    (4, 64, 64, 64, 64, 3, 1, 1),
    (2, 32, 32, 128, 256, 3, 2, 1),
    (2, 16, 16, 448, 128, 1, 1, 0),
    (1, 128, 128, 16, 16, 3, 1, 1),
    (2, 24, 40, 256, 16, 1, 1, 0),
    (4, 128, 128, 32, 128, 3, 1, 1),   # 512 tiles of 128x128
    (2, 64, 64, 16, 32, 3, 2, 1),      # DLA level1 shape (stride-2 dgrad on the 256x32 tile)
    (2, 32, 32, 64, 128, 1, 2, 0),     # ResNet downsample 1x1/s2
    (4, 128, 128, 128, 256, 3, 2, 1),  # stride-2 dgrad on 128x128 tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES)
def test_conv_gpu(hip_lib, case):
    _run_case("cuda", case)


@pytest.mark.gpu
def test_linear_gpu(hip_lib):
    _run_linear("cuda", 300, 12544, 1024)
    _run_linear("cuda", 130, 1024, 256)
    _run_linear("cuda", 1024, 2560, 1024)     # few 128x128 tiles + deep reduction: large tile with split-K (the fc1 shape class)
    # the head GEMMs at the benchmarked batch (4 images): cube head on 4 x 128 ROIs, box head on 4 x 512
    _run_linear("cuda", 512, 1024, 656)       # fused cube prediction heads (2+1+3+6+1) x 50 = 650 -> 656
    _run_linear("cuda", 512, 1024, 1024)      # cube fc2
    _run_linear("cuda", 512, 12544, 1024)     # cube fc1 (data gradient: transposed weights + the engine's NT form)
    _run_linear("cuda", 2048, 12544, 1024)    # box fc1
    _run_transposed_dgrad("cuda", 130, 1000, 4100)
    _run_fc1_class_gradients("cuda")                      # 8 x 32 = 256 tiles (dgrad), 4 x 32 = 128 tiles (wgrad): cut in 2
    _run_fc1_class_gradients("cuda", 2048, 12544, 1024)   # box fc1: 784 wgrad tiles = 3 whole + 16 tiles cut in 16; 1568 dgrad tiles
    _run_linear("cuda", 2048, 1024, 256)      # fused box predictor 51 + 200 -> 256


# ---- Winograd F(2x2, 3x3) path ------------------------------------------------------------------------
def _run_wino(dev, N, C, K, H, W, seed=3, tile=2):
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    dy = torch.randn(N, K, H, W, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.relu(F.conv2d(xr, wr, br, padding=1))
    ref.backward(dy)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
    xk, wk, dyk = cl(x), cl(w), cl(dy)
    y, V = wino.conv3x3_fwd(xk, wk, b.to(dev), relu=True, tile=tile)
    assert V.shape[0] == (tile + 2) ** 2
    scale = float(F.conv2d(x.abs(), w.abs(), None, padding=1).max())
    tol = 2e-6 if tile == 2 else 1e-5          # F(4x4,3x3) mixes magnitudes up to 25x in its transforms
    assert (y.cpu() - ref.detach()).abs().max() <= tol * scale
    dz = dyk * (ref.detach().to(dev) > 0)
    dx = wino.conv3x3_dgrad(cl(dz), wk, tile=tile)
    assert (dx.cpu() - xr.grad).abs().max() <= tol * float(F.conv_transpose2d(dy.abs(), w.abs(), padding=1).max())
    dw = wino.conv3x3_wgrad(V, cl(dz))
    assert (dw.cpu() - wr.grad).abs().max() <= 5 * tol * max(1.0, float(wr.grad.abs().max()))
    dx2, dw2 = wino.conv3x3_backward(V, cl(dz), wk, None)                       # fused dy transforms: same results
    # same arithmetic, separately compiled kernels (fp contraction may differ on the GPU) / split-K atomics for dw
    assert (dx2 - dx).abs().max() <= 1e-5 * max(1.0, float(dx.abs().max()))
    assert (dw2 - dw).abs().max() <= 1e-5 * max(1.0, float(dw.abs().max()))
    acc = torch.ones_like(wk)
    wino.conv3x3_wgrad(V, cl(dz), accum_into=acc)
    assert (acc.cpu() - 1.0 - wr.grad).abs().max() <= 5 * tol * max(1.0, float(wr.grad.abs().max()))


def test_winograd_emulated(emu_lib):
    _run_wino("cpu", 1, 8, 12, 6, 8)
    _run_wino("cpu", 2, 4, 4, 4, 4, seed=5)
    _run_wino("cpu", 1, 8, 12, 8, 12, tile=4)        # F(4x4, 3x3)
    _run_wino("cpu", 2, 4, 4, 4, 4, seed=5, tile=4)


def test_winograd_dispatch_rule():
    from omni3d_amd.kernels import wino
    assert wino.eligible((4, 256, 128, 128), (256, 256, 3, 3), 1, 1)
    assert wino.eligible((4, 256, 64, 64), (256, 256, 3, 3), 1, 1)
    assert wino.eligible((4, 512, 16, 16), (512, 512, 3, 3), 1, 1) and wino.dgrad_eligible((4, 512, 16, 16)) and not wino.dgrad_eligible((2, 512, 16, 16))
    assert not wino.eligible((4, 256, 8, 8), (256, 256, 3, 3), 1, 1)        # too few tiles
    assert wino.eligible((4, 64, 128, 128), (64, 64, 3, 3), 1, 1) and wino.tile_size((4, 64, 128, 128)) == 4
    assert not wino.eligible((4, 32, 128, 128), (32, 32, 3, 3), 1, 1)        # narrow
    assert wino.tile_size((4, 256, 128, 128)) == 4 and wino.tile_size((4, 256, 32, 32)) == 4       # F(4x4,3x3) from 256 tiles up
    assert wino.tile_size((4, 512, 16, 16)) == 2 and wino.tile_size((2, 256, 126, 128)) == 2
    assert not wino.eligible((4, 256, 128, 128), (256, 256, 3, 3), 2, 1)     # strided
    assert not wino.eligible((4, 256, 127, 128), (256, 256, 3, 3), 1, 1)     # odd extent


@pytest.mark.gpu
def test_winograd_gpu(hip_lib):
    _run_wino("cuda", 2, 128, 256, 64, 64)
    _run_wino("cuda", 1, 32, 64, 10, 6)
    _run_wino("cuda", 2, 128, 256, 64, 64, tile=4)
    _run_wino("cuda", 1, 32, 64, 12, 8, tile=4)


def _run_persistent_gemm(dev):
    """the persistent batched GEMM (prefetch carried across work items), forced onto a small problem: 16 x 2 items over
    8 workgroups = 4 items each, ragged M and N tiles"""
    from omni3d_amd import lib as L
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(8)
    V = torch.randn(16, 160, 64, generator=g).to(dev)
    U = torch.randn(16, 72, 64, generator=g).to(dev)
    out = wino.gemm_batched(V, U, algo=1, workgroups=8)
    ref = torch.einsum("bmc,bkc->bmk", V.cpu().double(), U.cpu().double())
    assert (out.cpu().double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    assert torch.equal(out, wino.gemm_batched(V, U))       # bit-identical to the one-tile-per-workgroup kernel (same k order)


def _run_prefetch_gemm(dev):
    """gemm_nt_pf_kernel (64x64 tiles, 2 / 4 slabs of buffer-load prefetch in flight): ragged M and N tiles, reduction depths of
    exactly one round of the prefetch ring (K = 64 with PF 2, K = 128 with PF 4) and of several; bit-identical to the classic
    double-buffered 64x64 kernel (same k order per accumulator)."""
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(9)
    for B, M, K, C in ((3, 150, 72, 64), (2, 70, 130, 128), (2, 129, 64, 192), (1, 64, 64, 512)):
        V = torch.randn(B, M, C, generator=g).to(dev)
        U = torch.randn(B, K, C, generator=g).to(dev)
        out = wino.gemm_batched(V, U, algo=4)
        ref = torch.einsum("bmc,bkc->bmk", V.cpu().double(), U.cpu().double())
        assert (out.cpu().double() - ref).abs().max() <= 1e-4 * ref.abs().max(), (B, M, K, C)
        assert torch.equal(out, wino.gemm_batched(V, U, algo=3)), (B, M, K, C)
        # the persistent form (gemm_nt_pfp_kernel, slab stream running through the tile boundaries): 8 workgroups = one per XCD
        # chunk walking several tiles each, 16 = two per chunk, 1024 = more workgroups than tiles (clamped); bit-identical
        for wgs in (8, 16, 1024):
            assert torch.equal(out, wino.gemm_batched(V, U, algo=5, workgroups=wgs)), (B, M, K, C, wgs)
    g5 = torch.Generator().manual_seed(10)
    V, U = torch.randn(5, 200, 128, generator=g5).to(dev), torch.randn(5, 100, 128, generator=g5).to(dev)   # 5 x 4 x 2 = 40 tiles: ragged chunks (40 % 8 == 0), K = PF slabs
    assert torch.equal(wino.gemm_batched(V, U, algo=4), wino.gemm_batched(V, U, algo=5, workgroups=8))
    V, U = torch.randn(3, 130, 256, generator=g5).to(dev), torch.randn(3, 64, 256, generator=g5).to(dev)    # 9 tiles over 8 chunks: one chunk holds two
    assert torch.equal(wino.gemm_batched(V, U, algo=4), wino.gemm_batched(V, U, algo=5, workgroups=8))
    assert torch.equal(wino.gemm_batched(V, U, algo=4), wino.gemm_batched(V, U, algo=5, workgroups=0))
    with pytest.raises(Exception):
        wino.gemm_batched(torch.randn(1, 64, 32).to(dev), torch.randn(1, 64, 32).to(dev), algo=4)     # C % 64 != 0: refused, not mangled
    with pytest.raises(Exception):
        wino.gemm_batched(V, U, algo=5, workgroups=12)                                                # not a multiple of 8
    # TN twin (Winograd-domain weight gradient): ragged row counts (a last slab of 6 rows, a split that gets fewer rows), ragged tiles
    for B, M, K, C in ((3, 70, 72, 64), (2, 300, 136, 68), (2, 1024, 64, 128), (1, 128, 64, 64)):
        V = torch.randn(B, M, C, generator=g).to(dev)
        dM = torch.randn(B, M, K, generator=g).to(dev)
        dU = wino.gemm_batched_wgrad(V, dM, algo=2)
        ref = torch.einsum("bmk,bmc->bkc", dM.cpu().double(), V.cpu().double())
        assert (dU.cpu().double() - ref).abs().max() <= 1e-4 * ref.abs().max(), (B, M, K, C)
        assert (dU - wino.gemm_batched_wgrad(V, dM, algo=1)).abs().max() <= 1e-5 * float(ref.abs().max())
    # several problems of different shapes in one launch (gemm_tn_multi_kernel): bit-identical to one launch each -- ordered split
    # reductions and atomic ones (single-split problems), item counts that are no multiple of 8, an empty problem
    from omni3d_amd.kernels import detmode
    shapes = ((3, 70, 72, 64), (2, 300, 136, 68), (2, 1024, 64, 128), (1, 128, 64, 64), (16, 600, 64, 64), (2, 0, 64, 64), (36, 40, 128, 64))
    probs = [(torch.randn(B, M, C, generator=g).to(dev), torch.randn(B, M, K, generator=g).to(dev)) for B, M, K, C in shapes]
    for flag in (True, False):
        prev = detmode.set_enabled(flag)
        try:
            multi = wino.gemm_batched_wgrad_multi(probs)
            for (V, dM), dU, shp in zip(probs, multi, shapes):
                one = wino.gemm_batched_wgrad(V, dM)
                if flag or shp[1] <= 256:
                    assert torch.equal(dU, one), (flag, shp)
                else:
                    assert (dU - one).abs().max() <= 2e-5 * float(one.abs().max()), (flag, shp)
        finally:
            detmode.set_enabled(prev)
    # the weight-gradient stream's context: the transforms back add into the gradient views after the common launch
    cshapes = ((16, 600, 64, 64), (36, 40, 128, 64), (16, 300, 72, 68))
    cprobs = [(torch.randn(B, M, C, generator=g).to(dev), torch.randn(B, M, K, generator=g).to(dev)) for B, M, K, C in cshapes]
    start = [torch.randn(K, C, 3, 3, generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _, _, K, C in cshapes]
    want, got = [], []
    for (V, dM), w0 in zip(cprobs, start):
        t = w0.clone(memory_format=torch.channels_last)
        wino.transform_dweights(wino.gemm_batched_wgrad(V, dM), t)
        want.append(t)
    with wino.batched_wgrads():
        for (V, dM), w0 in zip(cprobs, start):
            t = w0.clone(memory_format=torch.channels_last)
            assert wino.wgrad_into(V, dM, t) is None
            got.append(t)
        assert torch.equal(got[0], start[0])          # nothing has been launched yet
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    # two sources into one view (the RPN's shared convolution over the FPN levels), F(2x2) and F(4x4) mixed
    sh = ((16, 200, 64, 64), (36, 90, 64, 64), (16, 50, 64, 64))
    sp = [(torch.randn(B, M, C, generator=g).to(dev), torch.randn(B, M, K, generator=g).to(dev)) for B, M, K, C in sh]
    seq = start[0].clone(memory_format=torch.channels_last)
    for V, dM in sp:
        wino.transform_dweights(wino.gemm_batched_wgrad(V, dM), seq)
    tog = start[0].clone(memory_format=torch.channels_last)
    other = start[1].clone(memory_format=torch.channels_last)
    with wino.batched_wgrads():
        wino.wgrad_into(sp[0][0], sp[0][1], tog)
        wino.wgrad_into(cprobs[1][0], cprobs[1][1], other)
        wino.wgrad_into(sp[1][0], sp[1][1], tog)
        wino.wgrad_into(sp[2][0], sp[2][1], tog)
    assert torch.equal(seq, tog) and torch.equal(other, want[1])


def test_persistent_gemm_emulated(emu_lib):
    _run_persistent_gemm("cpu")


def test_prefetch_gemm_emulated(emu_lib):
    _run_prefetch_gemm("cpu")


@pytest.mark.gpu
def test_prefetch_gemm_gpu(hip_lib):
    _run_prefetch_gemm("cuda")


@pytest.mark.gpu
def test_persistent_gemm_gpu(hip_lib):
    _run_persistent_gemm("cuda")


# ---- direct stem convolution (csrc/stem_conv.hip) ----------------------------------------------------------
def _run_stem(dev, N, H, W, C, R, seed=4):
    from omni3d_amd.kernels import conv
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(16, C, R, R, generator=g) * 0.2
    ref = F.conv2d(x, w, None, padding=R // 2)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
    assert conv.stem_eligible(x.shape, w.shape, 1, R // 2)
    y = conv.stem_conv_fwd(cl(x), cl(w))
    scale = float(F.conv2d(x.abs(), w.abs(), None, padding=R // 2).max())
    assert y.shape == ref.shape and (y.cpu() - ref).abs().max() <= 2e-6 * scale
    dy = torch.randn(ref.shape, generator=g)
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=R // 2)
    dw = conv.stem_conv_wgrad(cl(x), cl(dy), R)
    tol = 2e-6 * float(torch.nn.grad.conv2d_weight(x.abs(), w.shape, dy.abs(), padding=R // 2).max())
    assert dw.shape == dw_ref.shape and (dw.cpu() - dw_ref).abs().max() <= tol
    acc = torch.ones_like(cl(w))
    conv.stem_conv_wgrad(cl(x), cl(dy), R, accum_into=acc)
    assert (acc.cpu() - 1.0 - dw_ref).abs().max() <= tol


def _run_stem_s2_wgrad(dev, N, H, W, seed=7):
    """the stride-2 member (DLA-34 level1, 16 -> 32): both reduction forms against torch's conv2d_weight"""
    from omni3d_amd.kernels import conv, detmode
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 16, H, W, generator=g)
    dy = torch.randn(N, 32, H // 2, W // 2, generator=g)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
    assert conv.stem_wgrad_eligible(x.shape, (32, 16, 3, 3), 2, 1) and not conv.stem_wgrad_eligible((N, 16, H + 1, W), (32, 16, 3, 3), 2, 1)
    dw_ref = torch.nn.grad.conv2d_weight(x, (32, 16, 3, 3), dy, stride=2, padding=1)
    tol = 2e-6 * float(torch.nn.grad.conv2d_weight(x.abs(), (32, 16, 3, 3), dy.abs(), stride=2, padding=1).max())
    for flag in (True, False):
        prev = detmode.set_enabled(flag)
        try:
            dw = conv.stem_conv_wgrad(cl(x), cl(dy), 3, stride=2)
            assert dw.shape == dw_ref.shape and (dw.cpu() - dw_ref).abs().max() <= tol, flag
            acc = torch.ones(32, 16, 3, 3).contiguous(memory_format=torch.channels_last).to(dev)
            conv.stem_conv_wgrad(cl(x), cl(dy), 3, accum_into=acc, stride=2)
            assert (acc.cpu() - 1.0 - dw_ref).abs().max() <= tol, flag
        finally:
            detmode.set_enabled(prev)


def _run_stem_dgrad(dev, N, H, W, stride, seed=9):
    """round 5: the stem data gradients (level0: rotated filter formed inside the kernel; level1: stride 2, parity classes in one
    launch) against torch's conv2d_input and against the implicit GEMM"""
    from omni3d_amd.kernels import conv
    g = torch.Generator().manual_seed(seed)
    K = 16 if stride == 1 else 32
    w = torch.randn(K, 16, 3, 3, generator=g) * 0.2
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(N, K, OH, OW, generator=g)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
    assert conv.stem_dgrad_eligible((N, 16, H, W), w.shape, stride, 1) and not conv.stem_dgrad_eligible((N, 16, H, W), w.shape, stride, 0)
    ref = torch.nn.grad.conv2d_input((N, 16, H, W), w, dy, stride=stride, padding=1)
    scale = float(torch.nn.grad.conv2d_input((N, 16, H, W), w.abs(), dy.abs(), stride=stride, padding=1).max())
    dx = conv.stem_conv_dgrad(cl(dy), cl(w), (H, W), stride)
    assert dx.shape == ref.shape and (dx.cpu() - ref).abs().max() <= 2e-6 * scale, (stride, float((dx.cpu() - ref).abs().max()), scale)
    gemm = conv.conv2d_dgrad(cl(dy), cl(w), (H, W), stride, 1)
    assert (dx - gemm).abs().max() <= 4e-6 * scale


def test_stem_conv_emulated(emu_lib):
    _run_stem("cpu", 1, 6, 70, 16, 3)      # ragged tile in x and y
    _run_stem("cpu", 2, 5, 9, 4, 7)
    _run_stem_s2_wgrad("cpu", 2, 10, 70)   # 5 x 35 outputs: ragged in both directions
    _run_stem_dgrad("cpu", 1, 6, 70, 1)
    _run_stem_dgrad("cpu", 2, 10, 70, 2)   # ragged 8 x 64 tiles
    _run_stem_dgrad("cpu", 1, 9, 67, 2)    # odd extents: the last row / column of dx meets a single tap


@pytest.mark.gpu
def test_stem_conv_gpu(hip_lib):
    _run_stem("cuda", 2, 128, 192, 16, 3)
    _run_stem("cuda", 2, 130, 100, 4, 7)
    _run_stem_s2_wgrad("cuda", 2, 132, 200)
    _run_stem_s2_wgrad("cuda", 4, 512, 512)       # level1 at the benchmark's size: more tiles than persistent workgroups
    _run_stem_dgrad("cuda", 2, 128, 192, 1)
    _run_stem_dgrad("cuda", 2, 132, 200, 2)
    _run_stem_dgrad("cuda", 1, 67, 131, 2)
    _run_stem_dgrad("cuda", 4, 512, 512, 2)       # the benchmark's level1


def _run_stem_autograd(dev):
    """functional.conv2d routes the level0 shape through the stem kernel (forward and data gradient)"""
    from omni3d_amd import functional as HF
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 16, 5, 66, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) * 0.2
    dy = torch.randn(1, 16, 5, 66, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(dy)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
    xk, wk = cl(x).requires_grad_(True), cl(w).requires_grad_(True)
    y = HF.conv2d(xk, wk, None, 1, 1)
    y.backward(cl(dy))
    assert (xk.grad.cpu() - xr.grad).abs().max() <= 2e-5 and (wk.grad.cpu() - wr.grad).abs().max() <= 2e-4


def test_stem_conv_autograd_emulated(emu_lib):
    _run_stem_autograd("cpu")


def _run_weights_multi(dev):
    """every filter transform of a pass in one launch == the per-filter launches (both tile sizes, U and U' in any combination)"""
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(9)
    items = []
    for (K, C, tile, want_u, want_flip) in ((32, 64, 2, True, True), (64, 32, 4, True, False), (48, 40, 4, True, True), (16, 16, 2, False, True),
                                            (128, 64, 4, True, True)):
        w = torch.randn(K, C, 3, 3, generator=g).contiguous(memory_format=torch.channels_last).to(dev)
        items.append((w, want_u, want_flip, tile))
    outs = wino.transform_weights_multi(items)
    for (w, want_u, want_flip, tile), (U, Uf) in zip(items, outs):
        U1, Uf1 = wino.transform_weights(w, want_u, want_flip, tile)
        assert (U is None) == (not want_u) and (Uf is None) == (not want_flip)
        if want_u:
            assert torch.equal(U, U1)
        if want_flip:
            assert torch.equal(Uf, Uf1)


def _run_weight_plan(dev):
    """wino_weight_scope(owner): the second pass transforms the filters of the first with one launch and produces the same outputs"""
    from omni3d_amd import functional as HF
    from omni3d_amd.cubercnn.modeling.layers import Conv2d
    torch.manual_seed(2)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = Conv2d(128, 128, kernel_size=3, padding=1), Conv2d(128, 128, kernel_size=3, padding=1)

        def forward(self, x):
            with HF.wino_weight_scope(self):
                return self.b(self.a(x, relu=True)) + self.a(x)          # `a` is used twice: transformed once
    net = Net().to(dev)
    x = torch.randn(2, 128, 32, 32).contiguous(memory_format=torch.channels_last).to(dev).requires_grad_(True)
    y1 = net(x)
    plan = net._omni_wino_plan_train
    assert len(plan) == 2 and all(flip for _, _, flip in plan)
    y1.square().mean().backward()
    g1 = [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]
    for p in net.parameters():
        p.grad = None
    x.grad = None
    calls = []
    real = HF.wino.transform_weights
    HF.wino.transform_weights = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y2 = net(x)
    finally:
        HF.wino.transform_weights = real
    assert not calls and torch.equal(y1, y2)                 # no per-filter launch on the second pass
    y2.square().mean().backward()
    for a, b in zip(g1, [p.grad for p in net.parameters()] + [x.grad]):
        assert (a - b).abs().max() <= 1e-5 * max(1.0, float(a.abs().max()))
    with torch.no_grad():
        net(x.detach())
    assert len(net._omni_wino_plan_infer) == 2 and not any(flip for _, _, flip in net._omni_wino_plan_infer)


def test_winograd_weights_multi_emulated(emu_lib):
    _run_weights_multi("cpu")
    _run_weight_plan("cpu")


@pytest.mark.gpu
def test_winograd_weights_multi_gpu(hip_lib):
    _run_weights_multi("cuda")
    _run_weight_plan("cuda")


def _run_bn_relu_wino_fusion(dev, shapes):
    """BasicBlock (conv -> BN -> ReLU -> 3x3 conv -> BN + residual + ReLU) with bn1 + ReLU applied inside conv2's Winograd input
    transform == the unfused sequence: outputs, input / parameter gradients, running statistics, batch counter"""
    import copy
    from omni3d_amd import functional as HF
    from omni3d_amd.cubercnn.modeling.backbone.dla import BasicBlock
    for (N, C, H, tile) in shapes:
        torch.manual_seed(C + H)
        blk = BasicBlock(C, C).to(dev).train()
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.3)
        x0 = torch.randn(N, C, H, H).contiguous(memory_format=torch.channels_last).to(dev)
        g = torch.randn(N, C, H, H).contiguous(memory_format=torch.channels_last).to(dev)
        assert HF.wino.eligible(x0.shape, blk.conv2.weight.shape, 1, 1) and HF.wino.tile_size(x0.shape) == tile
        res = {}
        for fused in (True, False):         # (the switch is off by default: measured slower on MI355X; the path stays tested)
            prev, HF._BN_WINO_FUSE = HF._BN_WINO_FUSE, fused
            try:
                b = copy.deepcopy(blk)
                x = x0.clone().requires_grad_(True)
                calls = []
                real = HF.bnpool.bn_finalize_fwd
                HF.bnpool.bn_finalize_fwd = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
                try:
                    y = b(x)
                finally:
                    HF.bnpool.bn_finalize_fwd = real
                assert bool(calls) == fused                      # the fused path really ran (or did not)
                y.backward(g)
                HF.side_join()
                res[fused] = [y.detach(), x.grad] + [p.grad for p in b.parameters()] + [b.bn1.running_mean, b.bn1.running_var,
                                                                                        b.bn1.num_batches_tracked.float()]
            finally:
                HF._BN_WINO_FUSE = prev
        for i, (a, c) in enumerate(zip(res[True], res[False])):
            assert (a - c).abs().max() <= 2e-5 * max(1.0, float(c.abs().max())), (C, H, i, float((a - c).abs().max()))


def test_bn_relu_winograd_fusion_emulated(emu_lib):
    _run_bn_relu_wino_fusion("cpu", [(4, 128, 16, 2), (1, 128, 64, 4)])


@pytest.mark.gpu
def test_bn_relu_winograd_fusion_gpu(hip_lib):
    _run_bn_relu_wino_fusion("cuda", [(4, 128, 64, 4), (4, 256, 32, 4), (4, 512, 16, 2), (4, 64, 128, 4)])


# ---- one 3x3 filter over several tensors through shared Winograd arrays (functional._WinoConv3x3Levels) -------------------
def _run_levels(dev, sizes, tile_expected, N=2, C=128, K=128, seed=3):
    """the RPN's shared convolution over FPN levels.  (1) without ReLU against torch's conv2d on every level: outputs, input gradients
    (level 0 has a second consumer), filter and bias gradients.  (2) with ReLU against the per-level path of the same kernels on the
    levels that take the same transform there: a ReLU mask flips wherever an output is within rounding of zero, so only two paths with
    the same arithmetic can be compared through it."""
    from omni3d_amd import functional as HF
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(seed)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
    xs = [torch.randn(N, C, h, w_, generator=g) for h, w_ in sizes]
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    b = torch.randn(K, generator=g) * 0.1
    dys = [torch.randn(N, K, h, w_, generator=g) for h, w_ in sizes]
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = [F.conv2d(x, wr, br, padding=1) for x in xr]
    torch.autograd.backward(ref + [(xr[0] * 0.5).sum()], dys + [torch.ones(())])                 # a second consumer of level 0
    xd = [cl(x).to(dev).requires_grad_(True) for x in xs]
    wd, bd = cl(w).to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    assert HF.conv3x3_levels_eligible(xd, wd) and wino.levels_tile([tuple(x.shape) for x in xd]) == tile_expected
    with HF.wino_weight_scope():
        ys = HF.conv3x3_levels(xd, wd, bd, relu=False)
    torch.autograd.backward(ys + [(xd[0] * 0.5).sum()], [cl(d).to(dev) for d in dys] + [torch.ones((), device=dev)])
    tol = 2e-3 if tile_expected == 4 else 5e-4
    for y, r in zip(ys, ref):
        assert (y.detach().cpu() - r.detach()).abs().max() <= tol * float(r.detach().abs().max())
    for a, r in zip(xd, xr):
        assert (a.grad.cpu() - r.grad).abs().max() <= tol * float(r.grad.abs().max())
    assert (wd.grad.cpu() - wr.grad).abs().max() <= tol * float(wr.grad.abs().max())
    assert (bd.grad.cpu() - br.grad).abs().max() <= 1e-4 * float(br.grad.abs().max())
    # (2) ReLU: the same tensors through both forms
    import contextlib
    xa = [cl(x).to(dev).requires_grad_(True) for x in xs]
    xb = [cl(x).to(dev).requires_grad_(True) for x in xs]
    with HF.wino_weight_scope():
        ya = HF.conv3x3_levels(xa, wd, bd, relu=True)
    with (wino.f22_only() if tile_expected == 2 else contextlib.nullcontext()), HF.wino_weight_scope():
        same = [l for l, x in enumerate(xb) if wino.eligible(tuple(x.shape), tuple(wd.shape), 1, 1) and wino.tile_size(tuple(x.shape)) == tile_expected]
        yb = {l: HF.conv2d(xb[l], wd, bd, 1, 1, relu=True) for l in same}
    assert same, "no level takes the same transform on the per-level path"
    torch.autograd.backward(ya, [cl(d).to(dev) for d in dys])
    torch.autograd.backward([yb[l] for l in same], [cl(dys[l]).to(dev) for l in same])
    for l in same:
        assert torch.equal(ya[l].detach(), yb[l].detach()), l                    # same arithmetic per output row
        assert (xa[l].grad - xb[l].grad).abs().max() <= 1e-5 * float(xb[l].grad.abs().max()), l


def test_wino_levels_emulated(emu_lib):
    _run_levels("cpu", [(16, 24), (8, 12), (4, 8)], 2, N=4)      # 128 tiles of 4x4 are below the 36-point transform's floor: 16-point
    _run_levels("cpu", [(32, 32), (16, 16), (8, 8), (4, 4)], 4, N=4)


@pytest.mark.gpu
def test_wino_levels_gpu(hip_lib):
    _run_levels("cuda", [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], 4, N=4, C=256, K=256)      # the benchmark's five levels
    _run_levels("cuda", [(64, 96), (32, 48), (16, 24), (8, 12), (4, 6)], 2, N=2, C=256, K=256)         # p6 of 4 x 6: 16-point transform


def test_collected_weight_gradients_emulated(emu_lib):
    """what solver/graphed.py does with a backward stage's weight-gradient closures, without the graphs: in side_mode("collect") the
    backward of Winograd / direct / linear layers only QUEUES its filter- and bias-gradient launches; run afterwards inside
    wino.batched_wgrads() (the Winograd-domain GEMMs leave in one launch, a filter shared by two layers receives both gradients in issue
    order) they must leave in the gradient views exactly what the inline mode leaves"""
    from omni3d_amd import functional as HF
    from omni3d_amd.kernels import wino
    g = torch.Generator().manual_seed(12)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731

    def build():
        ws = [cl(torch.randn(128, 128, 3, 3, generator=torch.Generator().manual_seed(s)) * 0.05).requires_grad_(True) for s in (1, 2)]
        bs = [torch.randn(128, generator=torch.Generator().manual_seed(9)).requires_grad_(True)]
        lw = (torch.randn(64, 128, generator=torch.Generator().manual_seed(7)) * 0.05).requires_grad_(True)
        for p in ws + bs + [lw]:
            p.grad = torch.zeros_like(p)              # the optimizer's bucket views: backward ADDS into them
            p._omni_direct_grad = True
        return ws, bs, lw

    xs = [cl(torch.randn(2, 128, 32, 16, generator=g)), cl(torch.randn(2, 128, 32, 32, generator=g))]       # 256 / 512 tiles: Winograd layers
    dys = [cl(torch.randn(2, 128, 32, 16, generator=g)), cl(torch.randn(2, 128, 32, 32, generator=g)), cl(torch.randn(2, 128, 32, 16, generator=g))]
    dl = torch.randn(2 * 32 * 16, 64, generator=g)
    assert all(wino.eligible(tuple(x.shape), (128, 128, 3, 3), 1, 1) for x in xs)

    def run(ws, bs, lw):
        xa, xb = xs[0].clone().requires_grad_(True), xs[1].clone().requires_grad_(True)
        with HF.wino_weight_scope():
            y0 = HF.conv2d(xa, ws[0], bs[0], 1, 1, relu=True)            # filter 0 + bias on two tensors (the RPN's shared conv)
            y1 = HF.conv2d(xb, ws[0], bs[0], 1, 1, relu=True)
            y2 = HF.conv2d(xa, ws[1], None, 1, 1)
            y3 = HF.linear(y2.permute(0, 2, 3, 1).reshape(-1, 128), lw, None)
        torch.autograd.backward([y0, y1, y2, y3], [dys[0], dys[1], dys[2], dl])
        return xa.grad, xb.grad

    prev = HF.side_mode()
    try:
        HF.side_mode("inline")
        ref_p = build()
        ref_dx = run(*ref_p)
        HF.side_mode("collect")
        got_p = build()
        got_dx = run(*got_p)
        fns, keep = HF.side_take()
        assert len(fns) >= 6                              # 3 filter gradients + 2 bias gradients + the linear layer's
        assert all(float(p.grad.abs().max()) == 0.0 for p in got_p[0] + got_p[1] + [got_p[2]]), "collect mode must not launch them"
        seen = []
        real = wino.gemm_batched_wgrad_multi
        wino.gemm_batched_wgrad_multi = lambda probs: (seen.append(len(probs)), real(probs))[1]
        try:
            with wino.batched_wgrads():
                for fn in fns:
                    fn()
        finally:
            wino.gemm_batched_wgrad_multi = real
        assert seen == [3], seen                          # the three Winograd layers' GEMMs left in ONE launch
    finally:
        HF.side_take()
        HF.side_mode(prev)
    for a, b in zip(ref_dx, got_dx):
        assert torch.equal(a, b)
    for a, b in zip(ref_p[0] + ref_p[1] + [ref_p[2]], got_p[0] + got_p[1] + [got_p[2]]):
        assert torch.equal(a.grad, b.grad)


def _run_multi_src(dev, cases):
    """round 5: the DLA Root's conv1x1(torch.cat(children, 1)) with the children read in place (omni_conv2d_fwd_multi_det) is
    BIT-identical to the single-tensor call on the concatenated copy -- output, BatchNorm statistics, every tile / split choice --
    and the autograd function hands back the same gradients as concatenate + conv2d"""
    from omni3d_amd import functional as HF
    from omni3d_amd.kernels import conv
    g = torch.Generator().manual_seed(21)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
    for (N, H, W, cs, K, tile, splits) in cases:
        xs = [cl(torch.randn(N, c, H, W, generator=g)) for c in cs]
        w = cl(torch.randn(K, sum(cs), 1, 1, generator=g) * 0.05)
        assert conv.multi_src_eligible(xs, w)
        cat = cl(torch.cat(xs, 1))
        y, parts = conv.conv1x1_multi_fwd(xs, w, want_stats=True, tile=tile, splits=splits)
        if tile == 0 and splits == 0:
            y0, parts0 = conv.conv2d_fwd_stats(cat, w, 1, 0)
            assert (parts is None) == (parts0 is None)
            if parts is not None:
                assert torch.equal(parts, parts0)
        else:
            y0 = conv.conv2d_fwd(cat, w, None, 1, 0, tile=tile, splits=splits)
        assert torch.equal(y, y0), (cs, tile, splits, float((y - y0).abs().max()))
        ref = F.conv2d(torch.cat([x.cpu() for x in xs], 1), w.cpu())
        assert (y.cpu() - ref).abs().max() <= 2e-5 * float(ref.abs().max())
        b = torch.randn(K, generator=g).to(dev)
        yb, _ = conv.conv1x1_multi_fwd(xs, w, bias=b, relu=True)
        assert torch.equal(yb, conv.conv2d_fwd(cat, w, b, 1, 0, relu=True))
        if tile == 0:       # the weight gradient reads the children in place as well
            dy = cl(torch.randn(N, K, H, W, generator=g))
            assert torch.equal(conv.conv1x1_multi_wgrad(xs, dy), conv.conv2d_wgrad(cat, dy, (1, 1), 1, 0))
            into = cl(torch.randn(K, sum(cs), 1, 1, generator=g))
            into2 = into.clone(memory_format=torch.channels_last)
            conv.conv1x1_multi_wgrad(xs, dy, accum_into=into)
            conv.conv2d_wgrad(cat, dy, (1, 1), 1, 0, accum_into=into2)
            assert torch.equal(into, into2)
    # the non-deterministic form (split reductions meet through atomics): same sums up to their order
    from omni3d_amd.kernels import detmode
    prev = detmode.set_enabled(False)
    try:
        N, H, W, cs, K = 2, 4, 4, (256, 256, 128, 256), 256
        xs = [cl(torch.randn(N, c, H, W, generator=g)) for c in cs]
        w = cl(torch.randn(K, sum(cs), 1, 1, generator=g) * 0.05)
        cat = cl(torch.cat(xs, 1))
        y, _ = conv.conv1x1_multi_fwd(xs, w, want_stats=True)
        y0 = conv.conv2d_fwd(cat, w, None, 1, 0)
        assert (y - y0).abs().max() <= 2e-5 * float(y0.abs().max())
        dy = cl(torch.randn(N, K, H, W, generator=g))
        d, d0 = conv.conv1x1_multi_wgrad(xs, dy), conv.conv2d_wgrad(cat, dy, (1, 1), 1, 0)
        assert (d - d0).abs().max() <= 2e-5 * float(d0.abs().max())
    finally:
        detmode.set_enabled(prev)
    # not served: a width that is not a multiple of 32, a 3 x 3 filter, a single input
    assert not conv.multi_src_eligible([xs[0], xs[0][:, :16]], torch.zeros(8, xs[0].shape[1] + 16, 1, 1))
    assert not conv.multi_src_eligible(xs, torch.zeros(8, sum(cs), 3, 3)) and not conv.multi_src_eligible(xs[:1], torch.zeros(8, cs[0], 1, 1))
    # autograd: gradients of the children (one of them with a second consumer: the fan-in slot) and of the filter
    N, H, W, cs, K = 2, 9, 11, (64, 32, 64), 48
    base = [torch.randn(N, c, H, W, generator=g) for c in cs]
    w0 = torch.randn(K, sum(cs), 1, 1, generator=g) * 0.05
    dy = cl(torch.randn(N, K, H, W, generator=g))
    grads = []
    for multi in (True, False):
        leaves = [cl(t.clone()).requires_grad_(True) for t in base]
        wk = cl(w0.clone()).requires_grad_(True)      # (clone: a 1 x 1 filter is channels_last as it is -- `cl` alone hands back w0 itself)
        xs = [HF.fanout(t * 1.0) if i == 0 else t * 1.0 for i, t in enumerate(leaves)]
        y = HF.cat_conv1x1(xs, wk) if multi else HF.conv2d(HF.cat_channels(xs), wk, None, 1, 0)
        extra = HF.conv2d(xs[0], cl(torch.ones(8, cs[0], 1, 1) * 0.01), None, 1, 0).sum()      # the second consumer of the first child
        (y * dy).sum().add(extra).backward()
        HF.side_join()
        grads.append([t.grad.clone() for t in leaves] + [wk.grad.clone()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    ref_leaves = [t.clone().requires_grad_(True) for t in base]
    wr = w0.clone().requires_grad_(True)
    (F.conv2d(torch.cat(ref_leaves, 1), wr) * dy.cpu()).sum().add(F.conv2d(ref_leaves[0], torch.ones(8, cs[0], 1, 1) * 0.01).sum()).backward()
    for a, r in zip(grads[0], [t.grad for t in ref_leaves] + [wr.grad]):
        assert (a.cpu() - r).abs().max() <= 1e-4 * max(float(r.abs().max()), 1.0)


MULTI_SRC_SMALL = [(2, 9, 11, (64, 64), 64, 0, 0), (1, 8, 8, (128, 128, 64, 128), 128, 0, 0), (2, 4, 4, (256, 256, 128, 256), 256, 0, 0),
                   (1, 6, 7, (32, 96, 32, 64, 32, 32), 40, 2, 3), (2, 5, 5, (64, 32), 72, 3, 1), (1, 16, 16, (64, 64), 128, 1, 2), (1, 5, 9, (32, 32), 16, 4, 1)]


def test_conv1x1_multi_source_emulated(emu_lib):
    _run_multi_src("cpu", MULTI_SRC_SMALL)


@pytest.mark.gpu
def test_conv1x1_multi_source_gpu(hip_lib):
    # the six Roots of DLA-34 at the benchmark's size (4 x 512 x 512) + the small cases
    _run_multi_src("cuda", [(4, 128, 128, (64, 64), 64, 0, 0), (4, 64, 64, (128, 128), 128, 0, 0), (4, 64, 64, (128, 128, 64, 128), 128, 0, 0),
                            (4, 32, 32, (256, 256), 256, 0, 0), (4, 32, 32, (256, 256, 128, 256), 256, 0, 0), (4, 16, 16, (512, 512, 256), 512, 0, 0)]
                   + MULTI_SRC_SMALL)


def _run_wgrad_batch(dev, big=False):
    """round 6: the direct weight gradients of a backward stage in one launch (conv_wgrad_multi_kernel) are bit-identical to their own
    launches -- 1 x 1, stride-2 3 x 3, the DLA Root's multi-source 1 x 1, mixed tile shapes, two problems adding into one view"""
    from omni3d_amd.kernels import conv, wino
    g = torch.Generator().manual_seed(21)
    CL = torch.channels_last

    def rnd(*shape):
        return torch.randn(*shape, generator=g).contiguous(memory_format=CL).to(dev)
    s = 4 if big else 1
    # (N, C, H, W, K, R, stride, pad)
    cases = [(2, 64, 16 * s, 16 * s, 128, 1, 1, 0), (2, 32, 16 * s, 16 * s, 64, 3, 2, 1), (1, 128, 8 * s, 8 * s, 256, 1, 1, 0),
             (2, 64, 12 * s, 12 * s, 64, 3, 2, 1), (2, 16, 16 * s, 16 * s, 32, 3, 2, 1), (1, 256, 8 * s, 8 * s, 256, 1, 1, 0)]
    probs = []
    for N, C, H, W, K, R, st, pad in cases:
        OH = (H + 2 * pad - R) // st + 1
        probs.append((rnd(N, C, H, W), rnd(N, K, OH, OH), (R, R), st, pad))
    xs = [rnd(2, 32, 8 * s, 8 * s), rnd(2, 64, 8 * s, 8 * s), rnd(2, 32, 8 * s, 8 * s)]
    dy_ms = rnd(2, 64, 8 * s, 8 * s)

    def grads():
        return ([torch.full((dy.shape[1], x.shape[1], k[0], k[1]), 0.5).contiguous(memory_format=CL).to(dev) for x, dy, k, _, _ in probs]
                + [torch.full((64, 128, 1, 1), 0.25).contiguous(memory_format=CL).to(dev)])

    def issue(into):
        for (x, dy, k, st, pad), gw in zip(probs, into):
            conv.conv2d_wgrad(x, dy, k, st, pad, accum_into=gw)
        conv.conv1x1_multi_wgrad(xs, dy_ms, accum_into=into[-1])
        conv.conv2d_wgrad(probs[0][0], probs[0][1], probs[0][2], probs[0][3], probs[0][4], accum_into=into[0])       # the same view once more
    one = grads()
    issue(one)                                   # every problem its own launch
    many = grads()
    prev, conv.WGRAD_BATCH = conv.WGRAD_BATCH, True          # (measured neutral in the step and left off by default: kernels/conv.py)
    with wino.batched_wgrads():                  # the weight-gradient stream's context: everything leaves when it closes
        issue(many)
        assert all(torch.equal(a, b) for a, b in zip(many, grads()))          # nothing launched yet
    conv.WGRAD_BATCH = prev
    for a, b in zip(one, many):
        assert torch.equal(a.cpu(), b.cpu())
    # and against the plain sum: dw of a fresh buffer
    ref0 = conv.conv2d_wgrad(probs[1][0], probs[1][1], probs[1][2], probs[1][3], probs[1][4])
    assert (many[1].cpu() - 0.5 - ref0.cpu()).abs().max() <= 1e-4 * float(ref0.abs().max())


def test_wgrad_batch_emulated(emu_lib):
    _run_wgrad_batch("cpu")


@pytest.mark.gpu
def test_wgrad_batch_gpu(hip_lib):
    _run_wgrad_batch("cuda")
    _run_wgrad_batch("cuda", big=True)


def _run_s2_dgrad(dev, cases):
    """round 6 (csrc/dgrad_s2.hip): the fused four-class data gradient of the 3x3 / stride-2 / pad-1 convolutions against autograd of
    F.conv2d (fp32 accumulation order differs: 1e-5 relative to the largest element), with and without a fan-in target, odd sizes,
    ragged edges, both channel-tile forms; two runs are bit-identical"""
    import torch.nn.functional as F
    from omni3d_amd.kernels import conv
    g = torch.Generator().manual_seed(33)
    CL = torch.channels_last
    for N, C, H, W, K in cases:
        x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
        w = (torch.randn(K, C, 3, 3, generator=g) * 0.1)
        y = F.conv2d(x, w, None, 2, 1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        ref = x.grad
        prev, conv._S2_DGRAD_MIN_WGS = conv._S2_DGRAD_MIN_WGS, 1
        prev32, conv._S2_DGRAD_C32 = conv._S2_DGRAD_C32, True
        try:
            dyd, wd = dy.contiguous(memory_format=CL).to(dev), w.contiguous(memory_format=CL).to(dev)
            got = conv.conv2d_dgrad(dyd, wd, (H, W), 2, 1)
            again = conv.conv2d_dgrad(dyd, wd, (H, W), 2, 1)
            carry = torch.randn(N, C, H, W, generator=g).contiguous(memory_format=CL).to(dev)
            base = carry.clone()
            out = conv.conv2d_dgrad(dyd, wd, (H, W), 2, 1, accum_into=carry)
        finally:
            conv._S2_DGRAD_MIN_WGS, conv._S2_DGRAD_C32 = prev, prev32
        tol = 2e-5 * max(float(ref.abs().max()), 1.0)
        assert (got.cpu() - ref).abs().max() <= tol, (N, C, H, W, K, float((got.cpu() - ref).abs().max()))
        assert torch.equal(got, again)
        assert out.data_ptr() == carry.data_ptr() and (out.cpu() - base.cpu() - ref).abs().max() <= tol
        # and the generic kernel (the A/B partner) agrees
        prev, conv._S2_DGRAD = conv._S2_DGRAD, False
        try:
            old = conv.conv2d_dgrad(dyd, wd, (H, W), 2, 1)
        finally:
            conv._S2_DGRAD = prev
        assert (old.cpu() - got.cpu()).abs().max() <= tol


def test_s2_dgrad_emulated(emu_lib):
    _run_s2_dgrad("cpu", [(1, 32, 16, 16, 32), (2, 64, 17, 15, 64), (1, 32, 30, 34, 64), (1, 128, 9, 16, 32)])


@pytest.mark.gpu
def test_s2_dgrad_gpu(hip_lib):
    # DLA-34's level 2 / level 3 entries at the benchmark's size, ResNet-34's stage entries, and the small ragged cases
    _run_s2_dgrad("cuda", [(4, 32, 256, 256, 64), (4, 64, 128, 128, 128), (4, 128, 64, 64, 256),
                           (1, 32, 16, 16, 32), (2, 64, 17, 15, 64), (1, 32, 30, 34, 64), (1, 128, 9, 16, 32)])
