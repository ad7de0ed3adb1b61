"""Gradient fan-in (functional.fanout): an activation with several consumers gets the SUM of their gradients, formed inside
the consumers' backward kernels (the *_carry entry points, omni_conv2d_dgrad(accumulate = 1)) instead of by autograd's add
kernels.  Checked here: every carry kernel against the plain kernel + a torch add (dense carries and channel slices of a wider
tensor, the layout of the DLA Root's concatenated gradient); pitched dy; the slot protocol on a small multi-consumer graph against
torch's own autograd of the same graph; the loud failure when a registered consumer never runs; the whole training step with the
switch on and off (GPU)."""
import pytest
import torch
import torch.nn.functional as F

CL = torch.channels_last


def _cl(t):
    return t.contiguous(memory_format=CL)


def _slice_of_wider(t, dev, lead=8, tail=4):
    """the same values as a channel slice [lead : lead + C] of a wider channels_last tensor (pixel pitch lead + C + tail)"""
    N, C, H, W = t.shape
    wide = _cl(torch.randn(N, lead + C + tail, H, W)).to(dev)
    wide[:, lead:lead + C] = t.to(dev)
    return wide[:, lead:lead + C]


def _run_carry_kernels(dev):
    from omni3d_amd import functional as HF
    from omni3d_amd.kernels import bnpool, conv, wino
    g = torch.Generator().manual_seed(11)
    for sliced in (False, True):
        mk = (lambda t: _slice_of_wider(t, dev)) if sliced else (lambda t: _cl(t).to(dev))
        # Winograd data gradient, both tile sizes
        for (N, C, K, H, W) in ((2, 32, 64, 8, 8), (1, 64, 32, 6, 10)):
            dy = torch.randn(N, K, H, W, generator=g)
            w = torch.randn(K, C, 3, 3, generator=g) * 0.1
            carry = torch.randn(N, C, H, W, generator=g)
            for tile in ((2, 4) if H % 4 == 0 and W % 4 == 0 else (2,)):
                plain = wino.conv3x3_dgrad(_cl(dy).to(dev), _cl(w).to(dev), tile=tile)
                c = mk(carry)
                assert HF._carry_pitch(c) == (c.stride(3))
                got = wino.conv3x3_dgrad(_cl(dy).to(dev), _cl(w).to(dev), tile=tile, carry=c)
                assert got.is_contiguous(memory_format=CL)
                assert torch.equal(got.cpu(), (plain + c).cpu()), ("wino", tile, sliced)
        # implicit-GEMM data gradient: in place on the carry, plain / stride 2 / split reduction
        for (N, C, K, H, W, R, stride, pad, splits) in ((2, 64, 32, 8, 8, 1, 1, 0, 0), (2, 32, 64, 8, 8, 3, 2, 1, 0), (1, 64, 512, 4, 4, 1, 1, 0, 2)):
            OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
            dy = torch.randn(N, K, OH, OW, generator=g)
            w = torch.randn(K, C, R, R, generator=g) * 0.1
            carry = torch.randn(N, C, H, W, generator=g)
            plain = conv.conv2d_dgrad(_cl(dy).to(dev), _cl(w).to(dev), (H, W), stride, pad)
            c = mk(carry)
            want = (plain + c).cpu()
            got = conv.conv2d_dgrad(_cl(dy).to(dev), _cl(w).to(dev), (H, W), stride, pad, splits=splits, accum_into=c)
            assert got.data_ptr() == c.data_ptr()
            assert (got.cpu() - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max())), ("dgrad", R, stride, splits, sliced)
        # BatchNorm backward: residual carry, pitched dy
        N, C, H, W = 2, 16, 6, 4
        x = torch.randn(N, C, H, W, generator=g)
        res = torch.randn(N, C, H, W, generator=g)
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        dy = torch.randn(N, C, H, W, generator=g)
        carry = torch.randn(N, C, H, W, generator=g)
        xk = _cl(x).to(dev)
        y, mean_rstd, _ = bnpool.bn_fwd(xk, gamma.to(dev), beta.to(dev), torch.zeros(C).to(dev), torch.ones(C).to(dev), residual=_cl(res).to(dev), relu=True)
        base = bnpool.bn_bwd(xk, _cl(dy).to(dev), y, gamma.to(dev), mean_rstd, relu=True, want_dres=True)
        got = bnpool.bn_bwd(xk, mk(dy), y, gamma.to(dev), mean_rstd, relu=True, want_dres=True, res_carry=mk(carry))
        assert torch.equal(got[0], base[0]) and torch.equal(got[2], base[2]) and torch.equal(got[3], base[3])
        assert torch.equal(got[1].cpu(), (base[1] + _cl(carry).to(dev)).cpu())
        got = bnpool.bn_bwd(xk, mk(dy), y, gamma.to(dev), mean_rstd, relu=True, want_dres=True)        # pitched dy alone
        assert all(torch.equal(a, b) for a, b in zip(got, base))
        # max-pool backward
        x = torch.randn(2, 8, 6, 10, generator=g)
        dy = torch.randn(2, 8, 3, 5, generator=g)
        carry = torch.randn(2, 8, 6, 10, generator=g)
        base = bnpool.maxpool2_bwd(_cl(x).to(dev), _cl(dy).to(dev))
        assert torch.equal(bnpool.maxpool2_bwd(_cl(x).to(dev), mk(dy), carry=mk(carry)).cpu(), (base + _cl(carry).to(dev)).cpu())
        assert torch.equal(bnpool.maxpool2_bwd(_cl(x).to(dev), mk(dy)), base)
        # FPN top-down backward, p6 subsampling backward
        dout = torch.randn(2, 8, 6, 4, generator=g)
        carry = torch.randn(2, 8, 3, 2, generator=g)
        base = bnpool.upsample2_bwd(_cl(dout).to(dev))
        assert torch.equal(bnpool.upsample2_bwd(_cl(dout).to(dev), carry=mk(carry)).cpu(), (base + _cl(carry).to(dev)).cpu())
        dy = torch.randn(2, 4, 4, 3, generator=g)
        carry = torch.randn(2, 4, 7, 5, generator=g)
        base = bnpool.subsample2_bwd(_cl(dy).to(dev), (7, 5))
        assert torch.equal(bnpool.subsample2_bwd(_cl(dy).to(dev), (7, 5), carry=mk(carry)).cpu(), (base + _cl(carry).to(dev)).cpu())
    # what the kernels cannot read as a carry is reported as such (the callers then add with torch)
    t = _cl(torch.randn(2, 8, 4, 4)).to(dev)
    assert HF._carry_pitch(t) == 8 and HF._carry_pitch(t[:, 2:6]) is None and HF._carry_pitch(t.contiguous()) is None


class _Net(torch.nn.Module):
    """a DLA-Tree-shaped toy: x feeds a max-pool and a stride-2 block; the block output feeds the next block, its residual and the
    Root's concatenation together with the pooled x"""

    def __init__(self, slots):
        super().__init__()
        from omni3d_amd.cubercnn.modeling.layers import BatchNorm2d, Conv2d
        self.slots = slots
        self.c1 = Conv2d(32, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.b1 = BatchNorm2d(64)
        self.c2 = Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=False)
        self.b2 = BatchNorm2d(64)
        self.root = Conv2d(64 + 64 + 32, 64, kernel_size=1, bias=False)
        self.top = Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, x):
        from omni3d_amd import functional as HF
        mark = HF.fanout if self.slots else (lambda t: t)
        mark(x)
        bottom = mark(HF.max_pool2(x))
        x1 = mark(self.b1(self.c1(x), relu=True))
        x2 = self.b2(self.c2(x1), residual=x1, relu=True)
        cat = HF.cat_channels((x2, x1, bottom)) if self.slots else torch.cat((x2, x1, bottom), 1)
        r = mark(self.root(cat))
        lat = torch.ones((r.shape[0], 64, 2 * r.shape[2], 2 * r.shape[3]), device=r.device).contiguous(memory_format=CL)
        up = HF.upsample2_add(lat, r)          # r: read by a 3x3 convolution, the FPN top-down sum and the p6 subsampling
        return self.top(r).square().mean() + up.square().mean() + HF.subsample2(r).sum() * 0.01


def _run_protocol(dev):
    from omni3d_amd import functional as HF
    torch.manual_seed(5)
    x0 = _cl(torch.randn(2, 32, 16, 16)).to(dev)
    grads = {}
    for slots in (True, False):
        torch.manual_seed(7)
        net = _Net(slots).to(dev).train()
        assert HF._FANOUT
        pre = _cl(torch.randn(32, 32, 1, 1) * 0.2).to(dev).requires_grad_(True)
        x = HF.conv2d(x0, pre, None, 1, 0)          # a non-leaf input: its gradient is the fan-in of the pool and the block
        net(x).backward()
        grads[slots] = [pre.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    for a, b in zip(grads[True], grads[False]):
        assert (a - b).abs().max() <= 2e-5 * max(1e-3, float(b.abs().max())), float((a - b).abs().max())


def _run_dead_consumer(dev):
    from omni3d_amd import functional as HF
    x0 = _cl(torch.randn(1, 32, 8, 8)).to(dev)
    w = _cl(torch.randn(32, 32, 1, 1)).to(dev).requires_grad_(True)
    x = HF.fanout(HF.conv2d(x0, w, None, 1, 0))
    live = HF.max_pool2(x)
    HF.max_pool2(x)                 # registers with the slot, its output is dropped
    with pytest.raises(RuntimeError, match="never ran in backward"):
        live.sum().backward()
    # with a gradient from outside the slot protocol the withheld sum is added to it instead
    x = HF.fanout(HF.conv2d(x0, w, None, 1, 0))
    live = HF.max_pool2(x)
    HF.max_pool2(x)
    w.grad = None
    (live.sum() + (x * 2.0).sum()).backward()
    want = torch.autograd.grad((F.max_pool2d(F.conv2d(x0, w), 2).sum() + (F.conv2d(x0, w) * 2.0).sum()), w)[0]
    assert (w.grad - want).abs().max() <= 1e-4 * float(want.abs().max())


def test_carry_kernels_emulated(emu_lib):
    _run_carry_kernels("cpu")


def test_slot_protocol_matches_autograd_emulated(emu_lib):
    _run_protocol("cpu")


def test_dead_consumer_is_loud_emulated(emu_lib):
    _run_dead_consumer("cpu")


@pytest.mark.gpu
def test_carry_kernels_gpu(hip_lib):
    _run_carry_kernels("cuda")


@pytest.mark.gpu
def test_slot_protocol_matches_autograd_gpu(hip_lib):
    _run_protocol("cuda")


@pytest.mark.gpu
def test_dead_consumer_is_loud_gpu(hip_lib):
    _run_dead_consumer("cuda")


def _step_grads(dev, name, fan):
    import os
    from conftest import ROOT
    from oracle import make_golden as MG
    from omni3d_amd import functional as HF, synthetic
    from omni3d_amd.d2.events import EventStorage
    prev, HF._FANOUT = HF._FANOUT, fan
    try:
        gold = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
        spec = gold["spec"]
        priors = synthetic.make_priors(50, bins=spec.get("prior_bins", 0))
        model = MG.build_product_model(MG.product_cfg(spec["overrides"], spec.get("config", "cubercnn_DLA34_FPN.yaml")), priors, spec["seed"], device=dev)
        batch = synthetic.make_batch(spec["images"], spec["height"], spec["width"], num_gt=spec["num_gt"], seed=spec["seed"], priors=priors)
        E = MG.variates(spec, gold["rpn_labels"].shape[1])
        model.proposal_generator.injected = {"E": E["rpn"], "proposals": gold["proposals"]}
        model.roi_heads.injected = {"E": E["roi"]}
        model.train()
        with EventStorage(0):
            losses = model(batch)
            sum(losses.values()).backward()
        return {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}, {k: float(v.detach()) for k, v in losses.items()}
    finally:
        HF._FANOUT = prev


def _run_step_on_off(dev, name):
    g1, l1 = _step_grads(dev, name, True)
    g0, l0 = _step_grads(dev, name, False)
    assert set(g0) == set(g1)
    if dev != "cuda":
        assert l0 == l1                            # the forward pass is untouched
        worst = max(float((g0[n] - g1[n]).norm()) / max(float(g0[n].norm()), 1e-12) for n in g0)
        assert worst < 2e-5, worst                 # the same sums in a different order: fp32 rounding, nothing more
        return worst
    # On the GPU two runs of the SAME configuration already differ (split-K atomics in forward and backward; the bottom-up's
    # gradient norms move by up to ~1 % from run to run, profiles/r03_grad_run_to_run_spread.txt), so the switch is held to the
    # parity caps of tests/test_model_parity.py between its two settings; the exact comparisons are the emulated test above and
    # the kernel / protocol tests of this file.
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    worst = 0.0
    for n in g0:
        a, b = float(g0[n].norm()), float(g1[n].norm())
        rel = abs(a - b) / max(a, 1e-6)
        worst = max(worst, rel)
        assert rel < (0.02 if "bottom_up" in n else 0.005) or abs(a - b) < 1e-6, (n, a, b)
    return worst


@pytest.mark.skipif(__import__("os").environ.get("OMNI_SLOW") != "1", reason="2.7 min under the host emulator (two emulated model steps); set OMNI_SLOW=1 "
                    "(the GPU variant is the gate, the fan-in kernels keep their own emulated tests above)")
def test_training_step_fanin_on_off_emulated(emu_lib):
    _run_step_on_off("cpu", "dla34_tiny")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dla34_small", "resnet34_small"])
def test_training_step_fanin_on_off_gpu(hip_lib, name):
    _run_step_on_off("cuda", name)
