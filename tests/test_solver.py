"""FlatSGD checkpoint interchange with torch.optim.SGD (the format the reference's DetectionCheckpointer writes,
tools/train_net.py:128), robustness against re-bound .grad tensors, and the StepGuard (tools/train_net.py:157-285)."""
import math
import os
import sys

import pytest
import torch
from torch import nn

from conftest import ROOT  # noqa: F401


def _nets():
    torch.manual_seed(0)
    a = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
    b = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
    b.load_state_dict(a.state_dict())
    return a, b


def _loss(net, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(7, 6, generator=g)
    return (net(x) ** 2).mean()


def _flat(net, **kw):
    from omni3d_amd.cubercnn.solver.build import FlatSGD
    return FlatSGD([{"params": [p], "lr": 0.1, "weight_decay": wd} for p, wd in zip(net.parameters(), (1e-2, 0.0, 1e-2, 0.0))],
                   0.1, momentum=0.9, direct_accumulate=False, **kw)


def _torch(net):
    return torch.optim.SGD([{"params": [p], "lr": 0.1, "weight_decay": wd} for p, wd in zip(net.parameters(), (1e-2, 0.0, 1e-2, 0.0))],
                           0.1, momentum=0.9)


def _same(a, b, tol=1e-6):
    for p, q in zip(a.parameters(), b.parameters()):
        assert (p - q).abs().max() <= tol, float((p - q).abs().max())


def test_flat_sgd_state_dict_is_torch_sgd_format(emu_lib):
    a, b = _nets()
    fo, to = _flat(a), _torch(b)
    for it in range(2):
        for net, opt in ((a, fo), (b, to)):
            opt.zero_grad()
            _loss(net, it).backward()
            opt.step()
    _same(a, b)
    sd = fo.state_dict()
    assert set(sd) == {"state", "param_groups"}
    assert all("momentum_buffer" in v for v in sd["state"].values()) and len(sd["state"]) == 4
    ref = to.state_dict()
    for k in ref["state"]:
        assert (sd["state"][k]["momentum_buffer"] - ref["state"][k]["momentum_buffer"]).abs().max() <= 1e-6
    # FlatSGD -> torch.optim.SGD and torch.optim.SGD -> FlatSGD, then one more step each: all four agree
    a2, b2 = _nets()
    a2.load_state_dict(a.state_dict()); b2.load_state_dict(b.state_dict())
    fo2, to2 = _flat(a2), _torch(b2)
    fo2.load_state_dict(to.state_dict())           # reference-format state into the flat optimizer (momentum must arrive)
    to2.load_state_dict(sd)                        # flat optimizer's state into torch SGD
    assert float(fo2.flat_mom.abs().sum()) > 0 and fo2._steps > 0
    for net, opt in ((a, fo), (b, to), (a2, fo2), (b2, to2)):
        opt.zero_grad()
        _loss(net, 9).backward()
        opt.step()
    _same(a, b); _same(a, a2); _same(a, b2)


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_lr_schedule_reaches_the_kernel_after_load_state_dict(emu_lib, kind):
    """torch's Optimizer.load_state_dict REPLACES the param_group dicts.  The fused step must read the live
    groups, else a resumed run (tools/train_net.py:128 resume_or_load, the divergence retry) trains at the constructor's lr for
    ever while the scheduler writes to dicts nobody reads."""
    from omni3d_amd.cubercnn.solver.build import FlatAdam
    a, b = _nets()
    if kind == "sgd":
        fo, to = _flat(a), _torch(b)
    else:
        groups = lambda net: [{"params": [p], "lr": 0.1, "weight_decay": wd} for p, wd in zip(net.parameters(), (1e-2, 0.0, 1e-2, 0.0))]  # noqa: E731
        fo, to = FlatAdam(groups(a), 0.1, eps=1e-2, direct_accumulate=False), torch.optim.Adam(groups(b), 0.1, eps=1e-2)
    for net, opt in ((a, fo), (b, to)):
        opt.zero_grad(); _loss(net, 0).backward(); opt.step()
    fo.load_state_dict(fo.state_dict())
    to.load_state_dict(to.state_dict())
    assert all(fo.param_groups[gi] is not None for _, _, gi in fo.segments)
    sched_f = torch.optim.lr_scheduler.LambdaLR(fo, lambda it: 0.01)       # what build_lr_scheduler does after a resume
    sched_t = torch.optim.lr_scheduler.LambdaLR(to, lambda it: 0.01)
    assert abs(fo.param_groups[0]["lr"] - 1e-3) < 1e-12
    before = [p.detach().clone() for p in a.parameters()]
    for net, opt in ((a, fo), (b, to)):
        opt.zero_grad(); _loss(net, 1).backward(); opt.step()
    sched_f.step(); sched_t.step()
    _same(a, b, 2e-6)
    # and the step really was a small-lr step: two orders of magnitude below what lr 0.1 would have moved
    moved = max(float((p - q).abs().max()) for p, q in zip(a.parameters(), before))
    assert 0 < moved < 5e-3, moved


def test_flat_sgd_rejects_partial_momentum_state(emu_lib):
    a, b = _nets()
    fo, to = _flat(a), _torch(b)
    to.zero_grad(); _loss(b, 0).backward(); to.step()
    sd = to.state_dict()
    sd["state"].pop(0)
    with pytest.raises(ValueError):
        fo.load_state_dict(sd)


def test_flat_sgd_survives_module_zero_grad(emu_lib):
    """nn.Module.zero_grad() sets .grad to None (torch >= 2): autograd then builds fresh gradient tensors outside the bucket.
    step() must pick them up instead of silently applying only weight decay and momentum."""
    a, b = _nets()
    fo, to = _flat(a), _torch(b)
    for it in range(3):
        a.zero_grad()               # NOT optimizer.zero_grad()
        b.zero_grad()
        _loss(a, it).backward()
        _loss(b, it).backward()
        fo.step(); to.step()
    _same(a, b)
    assert all(p.grad.data_ptr() == fo.flat_grad[fo._slot[id(p)][0]:].data_ptr() for p in a.parameters())


def test_flat_sgd_deferred_world_scale(emu_lib):
    a, b = _nets()
    fo, to = _flat(a), _torch(b)
    fo.zero_grad(); to.zero_grad()
    (2 * _loss(a, 0)).backward()        # "sum over 2 ranks"
    _loss(b, 0).backward()
    fo._grad_scale = 0.5                # what all_reduce_finish(defer_scale=True) leaves for the step kernel
    fo.step(); to.step()
    _same(a, b)
    assert fo._grad_scale == 1.0


def _reference_guard(seq, stabilize, period):
    """tools/train_net.py:157-285 restated on host floats: seq = [(total_loss, grads_nonfinite)] -> [(skipped, retry)]"""
    success = explode = 0
    recent = None
    out = []
    for total, bad in seq:
        if recent is None:
            recent = total * 2.0
        div = stabilize > 0 and (total > recent * 4.0 or not math.isfinite(total))
        if not div:
            recent = recent * (1 - 0.02) + total * 0.02
            if stabilize > 0 and bad:
                div = True
        if div:
            explode += 1
        else:
            success += 1
        tot = success + explode
        retry = (explode / tot) >= stabilize and tot > period * 0.5
        out.append((bool(div), bool(retry) and stabilize > 0))
    return out


def _run_guard(dev, stabilize):
    """omni_guard_pre / omni_guard_post (csrc/optim.hip) behind StepGuard.update against the loop's logic on host floats"""
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    seq = [(6.0, 0), (5.5, 0), (30.0, 0), (5.0, 1), (float("nan"), 0), (4.8, 0), (100.0, 0), (4.7, 0), (float("inf"), 0), (4.5, 0)]
    want = _reference_guard(seq, stabilize, 6)
    g = StepGuard(["b", "a"], stabilize, 6, dev)
    for (total, bad), (skip_w, retry_w) in zip(seq, want):
        g.nonfinite_flag[0] = float(bad)
        skipped, retry, red = g.update({"a": torch.tensor(total * 0.25, device=dev), "b": torch.tensor(total * 0.75, device=dev)})
        assert (skipped, retry) == (skip_w, retry_w), (total, bad, skipped, retry, skip_w, retry_w)
        assert float(g.skip) == float(skip_w)
        if math.isfinite(total):
            assert abs(red["total_loss"] - total) < 1e-5 and abs(red["a"] - total * 0.25) < 1e-5


@pytest.mark.parametrize("stabilize", [0.02, 0.5, 0.0])
def test_step_guard_matches_the_reference_logic(emu_lib, stabilize):
    _run_guard("cpu", stabilize)


@pytest.mark.gpu
@pytest.mark.parametrize("stabilize", [0.02, 0.5, 0.0])
def test_step_guard_matches_the_reference_logic_gpu(hip_lib, stabilize):
    """the a-19 device kernels on MI355X (tools/train_net.py:157-285)"""
    _run_guard("cuda", stabilize)


@pytest.mark.gpu
def test_step_guard_gates_the_fused_update_on_the_device_gpu(hip_lib):
    """sync=False, as bench.py / the graphed step run it: the host never reads the decision inside the step; the fused SGD kernel
    skips on the device flag (diverged loss, then non-finite gradients), and applies the update otherwise."""
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    a, b = _nets()
    a = a.to("cuda")
    fo, to = _flat(a), _torch(b)
    g = StepGuard(["l"], 0.5, 100, "cuda")
    fo.skip_flag = g.skip
    plan = [("ok", 1.0), ("diverged", 50.0), ("nan_grad", 1.0), ("ok", 1.0)]
    for it, (what, mult) in enumerate(plan):
        fo.zero_grad(); to.zero_grad()
        gen = torch.Generator().manual_seed(it)
        x = torch.randn(7, 6, generator=gen)
        la = (a(x.cuda()) ** 2).mean() * mult
        lb = (b(x) ** 2).mean() * mult
        la.backward(); lb.backward()
        if what == "nan_grad":
            fo.flat_grad[3] = float("nan")
        fo.check_nonfinite(g.nonfinite_flag)
        g.update({"l": la.detach()}, sync=False)
        fo.step()
        if what == "ok":
            to.step()
        torch.cuda.synchronize()
        assert float(g.skip) == (0.0 if what == "ok" else 1.0), (it, what)
    for p, q in zip(a.parameters(), b.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max() <= 2e-6


def _guard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omni3d_amd import lib as L
    L._install_for_tests(L.HipLibrary(os.path.join(ROOT, "tests", "hipemu", "libomni3d_emu.so"), emulated=True))   # host-emulated kernels
    from omni3d_amd.cubercnn.solver.guard import StepGuard, allreduce_dict
    g = StepGuard(["x", "y"], 0.5, 2, "cpu")
    res = []
    # step 0: sane; step 1: rank 1 alone sees non-finite gradients; step 2: rank 0 alone has a huge loss
    for step, (lx, bad) in enumerate([((1.0, 3.0), (0, 0)), ((1.0, 3.0), (0, 1)), ((400.0, 2.0), (0, 0))]):
        g.nonfinite_flag[0] = float(bad[rank])
        skipped, retry, red = g.update({"x": torch.tensor(lx[rank]), "y": torch.tensor(2.0)})
        res.append((skipped, retry, round(red["x"], 4), round(red["total_loss"], 4)))
    d = allreduce_dict({"k": torch.tensor(float(rank + 1)), "j": torch.tensor(10.0 * (rank + 1))})
    res.append((float(d["k"]), float(d["j"])))
    q.put((rank, res))
    dist.destroy_process_group()


def test_step_guard_one_collective_keeps_ranks_in_agreement(emu_lib):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400
    ps = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
    assert out[0] == out[1]                                   # identical decisions and reduced scalars on both ranks
    steps = out[0]
    assert steps[0][:2] == (False, False) and steps[0][2] == 2.0            # mean of x over ranks (1, 3)
    assert steps[1][0] is True                                # one rank's bad gradients skip the step everywhere
    assert steps[2][0] is True and steps[2][1] is True        # 2 of 3 exploded >= 0.5 and 3 > period / 2: retry
    assert steps[3] == (1.5, 15.0)                            # allreduce_dict average


# ---- SOLVER.TYPE adam / adam+amsgrad / adamw / adamw+amsgrad (cubercnn/solver/build.py:58-65) ---------------------------------------
ADAM_TYPES = ["adam", "adam+amsgrad", "adamw", "adamw+amsgrad"]
_WD = (1e-2, 0.0, 3e-2, 0.0)


def _adam_pair(kind, a, b, dev="cpu"):
    from omni3d_amd.cubercnn.solver.build import FlatAdam
    groups = lambda net: [{"params": [p], "lr": 0.05 if i % 2 else 0.02, "weight_decay": wd} for i, (p, wd) in enumerate(zip(net.parameters(), _WD))]  # noqa: E731
    ams, adamw = kind.endswith("+amsgrad"), kind.startswith("adamw")
    fo = FlatAdam(groups(a), 0.02, eps=1e-2, amsgrad=ams, decoupled=adamw, direct_accumulate=False)
    to = (torch.optim.AdamW if adamw else torch.optim.Adam)(groups(b), 0.02, eps=1e-2, amsgrad=ams)      # as the reference builds them
    return fo, to


def _run_adam(kind, dev):
    a, b = _nets()
    a = a.to(dev)
    fo, to = _adam_pair(kind, a, b, dev)
    for it in range(6):
        for net, opt, d in ((a, fo, dev), (b, to, "cpu")):
            opt.zero_grad()
            g = torch.Generator().manual_seed(it)
            x = torch.randn(7, 6, generator=g).to(d)
            ((net(x) ** 2).mean() * (10.0 if it == 3 else 1.0)).backward()
            opt.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max() <= 2e-6, (kind, float((p.detach().cpu() - q.detach()).abs().max()))
    return a, b, fo, to


@pytest.mark.parametrize("kind", ADAM_TYPES)
def test_flat_adam_matches_torch(emu_lib, kind):
    a, b, fo, to = _run_adam(kind, "cpu")
    # checkpoint interchange in both directions, then one more step each
    sd, ref = fo.state_dict(), to.state_dict()
    assert set(sd["state"][0]) == set(ref["state"][0]) and float(sd["state"][0]["step"]) == float(ref["state"][0]["step"]) == 6.0
    for k in ref["state"]:
        for name in ref["state"][k]:
            assert (sd["state"][k][name].float() - ref["state"][k][name].float()).abs().max() <= 2e-6, (k, name)
    a2, b2 = _nets()
    a2.load_state_dict(a.state_dict()); b2.load_state_dict(b.state_dict())
    fo2, to2 = _adam_pair(kind, a2, b2)
    fo2.load_state_dict(ref)
    to2.load_state_dict(sd)
    assert float(fo2.dev_step) == 6.0
    for net, opt in ((a, fo), (b, to), (a2, fo2), (b2, to2)):
        opt.zero_grad()
        _loss(net, 9).backward()
        opt.step()
    _same(a, b, 2e-6); _same(a, a2, 2e-6); _same(a, b2, 2e-6)


def test_flat_adam_skipped_step_keeps_the_bias_correction(emu_lib):
    """the guard's skip flag: no update, no step-count advance (the reference does not call step() on such an iteration)"""
    a, b = _nets()
    fo, to = _adam_pair("adam", a, b)
    fo.skip_flag = torch.zeros(1)
    for it in range(4):
        skip = it == 1
        fo.skip_flag.fill_(1.0 if skip else 0.0)
        for net, opt in ((a, fo), (b, to)):
            opt.zero_grad()
            _loss(net, it).backward()
            if not (skip and opt is to):
                opt.step()
    assert float(fo.dev_step) == 3.0
    _same(a, b, 2e-6)


@pytest.mark.parametrize("kind", ADAM_TYPES + ["sgd", "lamb"])
def test_build_optimizer_solver_types(emu_lib, kind):
    from omni3d_amd.cubercnn.config import get_cfg_defaults
    from omni3d_amd.cubercnn.solver import FlatAdam, FlatSGD, build_optimizer
    from omni3d_amd.d2.config import get_cfg
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    cfg.SOLVER.TYPE = kind
    net = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4))
    if kind == "lamb":
        with pytest.raises(ValueError, match="is not supported as an optimizer"):
            build_optimizer(cfg, net)
        return
    opt = build_optimizer(cfg, net)
    assert isinstance(opt, FlatSGD if kind == "sgd" else FlatAdam)
    if kind != "sgd":
        assert opt.decoupled == kind.startswith("adamw") and ("max_exp_avg_sq" in opt.flat_state) == kind.endswith("+amsgrad")
        assert all(g["eps"] == 1e-2 and g["betas"] == (0.9, 0.999) for g in opt.param_groups)
    wds = sorted({g["weight_decay"] for g in opt.param_groups})
    assert wds == sorted({cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM, cfg.SOLVER.WEIGHT_DECAY_BIAS} - {None})


@pytest.mark.gpu
def test_flat_adam_matches_torch_gpu(hip_lib):
    for kind in ADAM_TYPES:
        _run_adam(kind, "cuda")


# ---- fused parameter groups: several nn.Linear evaluated as one GEMM, laid out back to back in the optimizer's buckets ----------------
def _run_fused_groups(dev):
    """box predictor (cls_score + bbox_pred) and cube head (five bbox_3D_* heads): with the optimizer's layout the fused matrix and its
    gradient are bucket views -- same outputs, same parameter gradients and the same update as the concatenating path"""
    import copy
    from omni3d_amd import functional as HF
    from omni3d_amd.cubercnn.modeling.roi_heads.fast_rcnn import FastRCNNOutputs
    from omni3d_amd.cubercnn.solver.build import FlatSGD, fused_view, tag_fused_groups
    torch.manual_seed(4)
    head = FastRCNNOutputs(64, box2box_weights=(10.0, 10.0, 5.0, 5.0), num_classes=50).to(dev)
    for p in head.parameters():
        p.data.normal_(0, 0.1)
    ref = copy.deepcopy(head)
    x = torch.randn(96, 64, device=dev)
    g = torch.randn(96, head.fused_dim, device=dev)
    g[:, 5 * 50 + 1:] = 0.0                                   # the loss kernels write zeros into the padding columns
    # reference: concatenation + autograd + torch SGD
    yr = ref(x)
    yr.backward(g)
    opt_r = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    opt_r.step()
    # product: tagged layout
    tag_fused_groups(head)
    opt = FlatSGD([{"params": [p]} for p in head.parameters()], lr=0.1, momentum=0.9, weight_decay=1e-3)
    members = [head.cls_score.weight, head.bbox_pred.weight]
    fw = fused_view(members, True)
    assert fw is not None and fw[0].numel() == head.fused_dim * 64 and fw[0].data_ptr() == head.cls_score.weight.data_ptr()
    assert head.bbox_pred.weight.data_ptr() == fw[0].data_ptr() + 4 * 51 * 64            # back to back, no alignment gap
    assert float(fw[0][251 * 64:].abs().max()) == 0.0                                    # zero padding rows
    fb = fused_view([head.cls_score.bias, head.bbox_pred.bias], True)
    assert fb is not None and fb[0].numel() == head.fused_dim
    opt.zero_grad()
    y = head(x)
    assert (y - yr.detach()).abs().max() < 1e-4
    y.backward(g)
    HF.side_join()
    for (n, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and (p.grad - q.grad).abs().max() < 1e-3 * max(1.0, float(q.grad.abs().max())), n
    assert float(fw[1][251 * 64:].abs().max()) == 0.0                                    # nothing lands in the padding
    opt.step()
    for (n, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        assert (p.data - q.data).abs().max() < 1e-5, n
    # inference reads the fused matrix without gradient plumbing
    with torch.no_grad():
        assert (head(x) - ref(x)).abs().max() < 1e-4
    # a parameter that moved (model.to(), re-assigned .data) silently falls back to the concatenating path
    head.bbox_pred.weight.data = head.bbox_pred.weight.data.clone()
    assert fused_view(members, True) is None
    with torch.no_grad():
        assert (head(x) - ref(x)).abs().max() < 1e-4
    # without direct accumulation (a DistributedDataParallel reducer needs autograd's hooks) no gradient view is handed out
    head2 = copy.deepcopy(ref)
    tag_fused_groups(head2)
    opt2 = FlatSGD([{"params": [p]} for p in head2.parameters()], lr=0.1, direct_accumulate=False)
    assert fused_view([head2.cls_score.weight, head2.bbox_pred.weight], True) is None
    assert fused_view([head2.cls_score.weight, head2.bbox_pred.weight], False) is not None
    opt2.zero_grad()
    head2(x).backward(g)
    assert (head2.cls_score.weight.grad - (ref.cls_score.weight.grad)).abs().max() < 1e-3


def test_fused_parameter_groups_emulated(emu_lib):
    _run_fused_groups("cpu")


@pytest.mark.gpu
def test_fused_parameter_groups_gpu(hip_lib):
    _run_fused_groups("cuda")
