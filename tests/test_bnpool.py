"""BatchNorm / pooling / FPN top-down / preprocess kernels (csrc/bn_pool.hip) vs PyTorch fp32 CPU."""
import pytest
import torch
import torch.nn.functional as F


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _run_bn(dev, N, C, H, W, relu, with_res):
    from omni3d_amd.kernels import bnpool
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.7
    res = torch.randn(N, C, H, W, generator=g) if with_res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    dy = torch.randn(N, C, H, W, generator=g)
    # reference
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if with_res:
        z = z + rr
    yr = F.relu(z) if relu else z
    yr.backward(dy)
    # kernel
    rm_k, rv_k = rm.clone().to(dev), rv.clone().to(dev)
    xk = _cl(x).to(dev)
    y, mean_rstd, scale_shift = bnpool.bn_fwd(xk, gamma.to(dev), beta.to(dev), rm_k, rv_k,
                                              residual=_cl(res).to(dev) if with_res else None, relu=relu)
    assert (y.cpu() - yr.detach()).abs().max() < 2e-5
    assert (rm_k.cpu() - rm_ref).abs().max() < 1e-5 and (rv_k.cpu() - rv_ref).abs().max() < 1e-5
    dx, dres, dgamma, dbeta = bnpool.bn_bwd(xk, _cl(dy).to(dev), y, gamma.to(dev), mean_rstd, relu=relu, want_dres=with_res)
    assert (dx.cpu() - xr.grad).abs().max() < 5e-5
    assert (dgamma.cpu() - gr.grad).abs().max() < 2e-4 * max(1.0, gr.grad.abs().max().item())
    assert (dbeta.cpu() - br.grad).abs().max() < 2e-4 * max(1.0, br.grad.abs().max().item())
    if with_res:
        assert (dres.cpu() - rr.grad).abs().max() < 1e-6
    if relu and not with_res:
        # the ReLU mask recomputed from x and the forward pass's (scale, shift) instead of read from y: the very same gradients
        dx2, _, dgamma2, dbeta2 = bnpool.bn_bwd(xk, _cl(dy).to(dev), None, gamma.to(dev), mean_rstd, relu=True, scale_shift=scale_shift)
        assert torch.equal(dx2, dx) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta)


def _run_pool(dev):
    from omni3d_amd.kernels import bnpool
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 6, 10, generator=g)
    x[0, :, 0, 0] = x[0, :, 0, 1]  # a tie: first maximum must win
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xk = _cl(x).to(dev)
    assert torch.equal(bnpool.maxpool2_fwd(xk).cpu(), yr.detach())
    assert torch.equal(bnpool.maxpool2_bwd(xk, _cl(dy).to(dev)).cpu(), xr.grad)
    # stride-2 subsample, odd size
    x = torch.randn(2, 4, 7, 5, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, kernel_size=1, stride=2, padding=0)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    assert torch.equal(bnpool.subsample2_fwd(_cl(x).to(dev)).cpu(), yr.detach())
    assert torch.equal(bnpool.subsample2_bwd(_cl(dy).to(dev), (7, 5)).cpu(), xr.grad)
    # FPN top-down
    lat, top = torch.randn(2, 8, 6, 4, generator=g), torch.randn(2, 8, 3, 2, generator=g)
    tr = top.clone().requires_grad_(True)
    out = lat + F.interpolate(tr, scale_factor=2.0, mode="nearest")
    dout = torch.randn_like(out)
    out.backward(dout)
    assert torch.equal(bnpool.upsample2_add(_cl(lat).to(dev), _cl(top).to(dev)).cpu(), out.detach())
    assert (bnpool.upsample2_bwd(_cl(dout).to(dev)).cpu() - tr.grad).abs().max() < 1e-6
    # preprocess
    img = torch.randint(0, 256, (2, 3, 50, 70), generator=g, dtype=torch.uint8)
    mean, std = [103.530, 116.280, 123.675], [57.375, 57.120, 58.395]
    got = bnpool.preprocess(img.to(dev), mean, std, 64).cpu()
    ref = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    assert got.shape == (2, 4, 64, 128)
    assert torch.equal(got[:, :3, :50, :70], ref)
    assert got[:, 3].abs().max() == 0 and got[:, :, 50:].abs().max() == 0 and got[:, :, :, 70:].abs().max() == 0


@pytest.mark.parametrize("cfg", [(2, 16, 5, 7, True, False), (1, 64, 4, 4, True, True), (3, 8, 3, 3, False, False),
                                 (2, 512, 2, 2, False, True),
                                 (1, 8, 384, 384, True, False),      # 72 partial blocks: the unrolled partial-sum loop
                                 (2, 1152, 3, 3, True, True), (1, 2048, 2, 4, False, False)])   # > 1024 channels: several 256-quad tiles
def test_bn_emulated(emu_lib, cfg):
    _run_bn("cpu", *cfg)


def test_pool_emulated(emu_lib):
    _run_pool("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(4, 16, 64, 64, True, False), (4, 128, 32, 32, True, True), (2, 512, 16, 16, False, False),
                                 (2, 1024, 4, 4, True, True), (4, 16, 512, 512, True, False), (4, 64, 128, 128, True, True)])
def test_bn_gpu(hip_lib, cfg):
    _run_bn("cuda", *cfg)


@pytest.mark.gpu
def test_pool_gpu(hip_lib):
    _run_pool("cuda")


def _run_pool3(dev):
    from omni3d_amd.kernels import bnpool
    g = torch.Generator().manual_seed(12)
    for shape in ((2, 8, 9, 12), (1, 4, 6, 6)):
        x = torch.randn(*shape, generator=g)
        x[0, :, 0, 0] = x[0, :, 1, 1] = 5.0      # tie inside the first window: first maximum wins
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(xr, 3, 2, 1)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        xk = _cl(x).to(dev)
        assert torch.equal(bnpool.maxpool3s2_fwd(xk).cpu(), yr.detach())
        assert (bnpool.maxpool3s2_bwd(xk, _cl(dy).to(dev)).cpu() - xr.grad).abs().max() < 1e-6


def test_pool3_emulated(emu_lib):
    _run_pool3("cpu")


@pytest.mark.gpu
def test_pool3_gpu(hip_lib):
    _run_pool3("cuda")


# ---- BatchNorm statistics emitted by the producing kernel's epilogue (conv / Winograd / stem) ----------------------------
def _bn_with_and_without_partials(dev, x_shape, w_shape, stride, pad):
    """conv -> training BatchNorm through omni3d_amd.functional with the fused statistics vs the separate statistics pass"""
    import omni3d_amd.functional as HF
    from omni3d_amd.kernels import bnpool
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*x_shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(*w_shape, generator=g) * 0.2).to(dev).contiguous(memory_format=torch.channels_last)
    C = w_shape[0]
    gamma, beta = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev)
    y = HF.conv2d(x, w, None, stride, pad, False, True)
    parts = getattr(y, "_omni_bn_partials", None)
    rm1, rv1, rm2, rv2 = [torch.zeros(C, device=dev) if i % 2 == 0 else torch.ones(C, device=dev) for i in range(4)]
    a, ms_a, _ = bnpool.bn_fwd(y, gamma, beta, rm1, rv1, None, True, 1e-5, 0.1, parts)
    b, ms_b, _ = bnpool.bn_fwd(y, gamma, beta, rm2, rv2, None, True, 1e-5, 0.1, None)
    return parts, (a - b).abs().max().item(), (ms_a - ms_b).abs().max().item() / ms_b.abs().max().item(), (rv1 - rv2).abs().max().item()


STATS_CASES = [
    ((2, 16, 24, 20), (32, 16, 3, 3), 1, 1, True),      # direct conv, 256x32 tile (K <= 32), ragged last m-tile
    ((2, 32, 20, 20), (72, 32, 1, 1), 1, 0, True),      # 1x1, 64x64 / 128x128 tiles with a ragged channel tile
    ((1, 8, 18, 14), (48, 8, 3, 3), 2, 1, True),        # stride 2
    ((1, 64, 6, 6), (64, 64, 3, 3), 1, 1, "split"),     # few tiles + deep reduction: split-K chosen -> statistics only from the ordered
                                                        # (deterministic) split, whose last-arriving workgroup holds the complete tile
    ((1, 4, 12, 70), (16, 4, 7, 7), 1, 3, True),        # stem kernel 7x7 4 -> 16
    ((1, 16, 9, 66), (16, 16, 3, 3), 1, 1, True),       # stem kernel 3x3 16 -> 16
    ((1, 128, 32, 32), (128, 128, 3, 3), 1, 1, True),   # Winograd F(2x2,3x3) output transform (256 tiles)
]


@pytest.mark.parametrize("case", STATS_CASES)
def test_bn_statistics_from_producer_epilogue_emulated(emu_lib, case):
    xs, ws, st, pad, expect = case
    from omni3d_amd.kernels import detmode
    parts, dy, dms, drv = _bn_with_and_without_partials("cpu", xs, ws, st, pad)
    assert (parts is not None) == (detmode.on() if expect == "split" else expect)
    if parts is not None:
        assert dy <= 2e-5 and dms <= 1e-5 and drv <= 1e-5, (dy, dms, drv)


@pytest.mark.gpu
def test_bn_statistics_from_producer_epilogue_gpu(hip_lib):
    for xs, ws, st, pad, _ in STATS_CASES + [((4, 256, 128, 128), (256, 256, 3, 3), 1, 1, True), ((4, 16, 512, 512), (16, 16, 3, 3), 1, 1, True),
                                             ((4, 64, 128, 128), (128, 64, 3, 3), 2, 1, True)]:
        parts, dy, dms, drv = _bn_with_and_without_partials("cuda", xs, ws, st, pad)
        if parts is not None:
            assert dy <= 5e-5 and dms <= 1e-5 and drv <= 1e-5, (xs, ws, dy, dms, drv)


@pytest.mark.gpu
def test_bias_grad_accumulates_with_atomics(hip_lib):
    """small tensors: one launch, column sums added to the running gradient with float atomics (csrc/bn_pool.hip)"""
    from omni3d_amd.kernels import bnpool
    g = torch.Generator().manual_seed(1)
    for shape in ((4, 128, 32, 32), (4, 256, 128, 128)):        # atomic path / partial rows + finalize
        dy = _cl(torch.randn(*shape, generator=g)).cuda()
        dy2 = dy.permute(0, 2, 3, 1).reshape(-1, shape[1])
        acc = torch.zeros(shape[1], device="cuda")
        for _ in range(3):
            bnpool.bias_grad(dy2, accum_into=acc)
        want = dy2.double().sum(0) * 3
        assert (acc.double() - want).abs().max() < 1e-3 * max(1.0, want.abs().max().item())


def _run_bn_bwd_stats_from_dgrad(dev, relu):
    """conv -> BatchNorm(+ReLU) -> 3x3 conv: the Winograd data-gradient transform of the second convolution leaves the BatchNorm's
    backward reductions behind (functional._BatchNorm.backward must pick them up) -- same gradients as the three-kernel backward"""
    from omni3d_amd import functional as HF
    from omni3d_amd.kernels import bnpool, wino
    g = torch.Generator().manual_seed(8)
    N, C, H, W = 4, 128, 32, 32          # a DLA level-4-like block: 256 tiles of 4 x 4, Winograd F(4x4,3x3) forward and data gradient
    x0 = torch.randn(N, C, H, W, generator=g).contiguous(memory_format=torch.channels_last).to(dev)
    w1 = (torch.randn(C, C, 3, 3, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    w2 = (torch.randn(C, C, 3, 3, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    dout = torch.randn(N, C, H, W, generator=g).contiguous(memory_format=torch.channels_last).to(dev)
    assert wino.eligible((N, C, H, W), (C, C, 3, 3), 1, 1) and wino.dgrad_eligible((N, C, H, W)) and wino.tile_size((N, C, H, W)) == 4
    taken = []
    real = bnpool.bn_bwd

    def spy(*a, **kw):
        taken.append(kw.get("partials") is not None)
        return real(*a, **kw)

    def run(fuse):
        HF._BN_BWD_FUSE = fuse
        xs = [t.clone().requires_grad_(True) for t in (x0, w1, w2, gamma, beta)]
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y = HF.batch_norm_train(HF.conv2d(xs[0], xs[1], None, 1, 1, False, True), xs[3], xs[4], rm, rv, relu=relu)
        out = HF.conv2d(y, xs[2], None, 1, 1)
        bnpool.bn_bwd = spy
        try:
            (out * dout).sum().backward()
        finally:
            bnpool.bn_bwd = real
        return [t.grad.clone() for t in xs]
    prev = HF._BN_BWD_FUSE
    try:
        ga = run(True)
        assert taken == [True], taken
        gb = run(False)
        assert taken == [True, False], taken
    finally:
        HF._BN_BWD_FUSE = prev
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("relu", [True, False])
def test_bn_backward_statistics_from_dgrad_emulated(emu_lib, relu):
    _run_bn_bwd_stats_from_dgrad("cpu", relu)


@pytest.mark.gpu
@pytest.mark.parametrize("relu", [True, False])
def test_bn_backward_statistics_from_dgrad_gpu(hip_lib, relu):
    _run_bn_bwd_stats_from_dgrad("cuda", relu)


def test_eval_coefficients_follow_the_running_statistics_emulated(emu_lib):
    """layers.BatchNorm2d caches (scale, shift) of the frozen statistics in eval mode; a training-mode pass in between moves the running
    statistics through the kernel (raw pointers, no version bump) and must invalidate the cache"""
    from omni3d_amd.cubercnn.modeling.layers import BatchNorm2d
    g = torch.Generator().manual_seed(2)
    bn = BatchNorm2d(8)
    x = torch.randn(2, 8, 4, 6, generator=g).contiguous(memory_format=torch.channels_last)
    ref = torch.nn.BatchNorm2d(8)
    ref.load_state_dict(bn.state_dict())
    bn.eval(); ref.eval()
    assert (bn(x) - ref(x)).abs().max() <= 1e-6
    first = bn.__dict__["_eval_scale_shift"][1]
    assert bn(x) is not None and bn.__dict__["_eval_scale_shift"][1] is first            # cached
    bn.train(); ref.train()
    bn(x * 3.0 + 1.0); ref(x * 3.0 + 1.0)                                                # moves running_mean / running_var
    bn.eval(); ref.eval()
    assert (bn.running_var - ref.running_var).abs().max() <= 1e-5
    assert (bn(x) - ref(x)).abs().max() <= 1e-5
    assert bn.__dict__["_eval_scale_shift"][1] is not first
    with torch.no_grad():
        bn.weight.mul_(2.0); ref.weight.mul_(2.0)                                        # a write torch sees
    assert (bn(x) - ref(x)).abs().max() <= 1e-5


def _fused_vs_three_launch(dev, N, C, H, W, relu, with_res, pitched):
    """round 5: the finalize folded into the apply launch (omni_bn_fwd_algo / omni_bn_bwd_algo, fuse_rows > 0) against the separate
    finalize launch (fuse_rows = 0) -- same summation order per channel, so every output must be the same BITS; `pitched`: y is written
    into / dy is read from a channel slice of a wider NHWC tensor (the DLA Root's concatenated input, dla.py:171)"""
    from omni3d_amd.kernels import bnpool
    g = torch.Generator().manual_seed(N * 1000 + C + H)
    x = _cl(torch.randn(N, C, H, W, generator=g) * 2 + 0.7).to(dev)
    res = _cl(torch.randn(N, C, H, W, generator=g)).to(dev) if with_res else None
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    dy = _cl(torch.randn(N, C, H, W, generator=g)).to(dev)
    carry = _cl(torch.randn(N, C, H, W, generator=g)).to(dev) if with_res else None
    wide = C + 32
    outs = []
    for rows in (0, 4096):
        bnpool.FUSE_ROWS = rows
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        out = None
        if pitched:
            big = torch.full((N, H, W, wide), 7.0, device=dev)
            out = big.permute(0, 3, 1, 2)[:, 16:16 + C]
        y, mean_rstd, scale_shift = bnpool.bn_fwd(x, gamma, beta, rm, rv, residual=res, relu=relu, out=out)
        if pitched:
            assert y.data_ptr() == out.data_ptr()
            assert (big[..., :16] == 7.0).all() and (big[..., 16 + C:] == 7.0).all()
            dyk = torch.zeros((N, H, W, wide), device=dev)
            dyk[..., 8:8 + C] = dy.permute(0, 2, 3, 1)
            dyk = dyk.permute(0, 3, 1, 2)[:, 8:8 + C]
        else:
            dyk = dy
        remask = relu and not with_res
        dx, dres, dgamma, dbeta = bnpool.bn_bwd(x, dyk, None if remask else (y.contiguous(memory_format=torch.channels_last) if relu else None),
                                                gamma, mean_rstd, relu=relu, want_dres=with_res, scale_shift=scale_shift if remask else None,
                                                res_carry=carry)
        outs.append([y.clone(), mean_rstd, scale_shift, rm, rv, dx, dgamma, dbeta] + ([dres] if with_res else []))
    bnpool.FUSE_ROWS = 512
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # and the values themselves against torch
    xr = x.cpu().clone().requires_grad_(True)
    rr = res.cpu().clone().requires_grad_(True) if with_res else None
    z = F.batch_norm(xr, torch.zeros(C), torch.ones(C), gamma.cpu(), beta.cpu(), True, 0.1, 1e-5)
    z = z + rr if with_res else z
    yr = F.relu(z) if relu else z
    yr.backward(dy.cpu())
    assert (outs[1][0].cpu() - yr.detach()).abs().max() < 2e-5
    assert (outs[1][5].cpu() - xr.grad).abs().max() < 5e-5
    if with_res:
        assert (outs[1][8].cpu() - (rr.grad + carry.cpu())).abs().max() < 1e-6


FUSED_CASES = [(2, 16, 5, 7, True, False, False), (1, 64, 4, 4, True, True, True), (2, 32, 9, 9, False, False, True),
               (1, 16, 70, 66, True, False, False),     # several pixel chunks with a ragged tail, > 256 partial rows in backward
               (2, 48, 6, 10, True, True, False), (1, 128, 3, 5, False, True, True)]


@pytest.mark.parametrize("cfg", FUSED_CASES)
def test_bn_finalize_folded_into_apply_emulated(emu_lib, cfg):
    _fused_vs_three_launch("cpu", *cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", FUSED_CASES + [(4, 128, 64, 64, True, True, False), (4, 256, 32, 32, True, False, True),
                                               (4, 512, 16, 16, True, True, True), (4, 64, 128, 128, True, False, False)])
def test_bn_finalize_folded_into_apply_gpu(hip_lib, cfg):
    _fused_vs_three_launch("cuda", *cfg)
