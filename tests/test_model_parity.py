"""End-to-end parity of the HIP Cube R-CNN training step against golden fixtures produced by the
REFERENCE's own files run on CPU (oracle/make_golden.py -> tests/golden/dla34_small.pt):
anchor labels and sampled ROI classes exactly, the 10 losses and parameter gradients within fp32
tolerance.  Weights come from the same CPU seed on both sides; sampling variates are injected."""
import os

import pytest
import torch

from conftest import ROOT

def _gold(name):
    return os.path.join(ROOT, "tests", "golden", name + ".pt")


def _run(dev, name):
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    from omni3d_amd.d2.events import EventStorage
    gold = torch.load(_gold(name), weights_only=False)
    spec = gold["spec"]
    priors = synthetic.make_priors(50)
    model = MG.build_product_model(MG.product_cfg(spec["overrides"], spec.get("config", "cubercnn_DLA34_FPN.yaml")), priors, spec["seed"], device=dev)
    batch = synthetic.make_batch(spec["images"], spec["height"], spec["width"], num_gt=spec["num_gt"], seed=spec["seed"], priors=priors)
    A = gold["rpn_labels"].shape[1]
    E = MG.variates(spec, A)
    model.proposal_generator.injected = {"E": E["rpn"]}
    model.roi_heads.injected = {"E": E["roi"]}
    model.train()
    with EventStorage(0) as st:
        losses = model(batch)
        sum(losses.values()).backward()
        logs = {k: v[0] for k, v in st.latest().items()}
    # integer decisions: exact
    assert torch.equal(model.proposal_generator.last_labels.cpu(), gold["rpn_labels"])
    for name in ("rpn/num_pos_anchors", "rpn/num_neg_anchors", "roi_head/num_fg_samples", "roi_head/num_bg_samples"):
        assert abs(logs[name] - gold["logs"][name]) < 1e-6, name
    # losses: fp32 tolerance (north star: 1e-4 on box params; losses are O(1))
    for k, v in gold["losses"].items():
        got = float(losses[k])
        assert abs(got - v) <= 2e-4 * max(1.0, abs(v)), (k, got, v)
    for name in ("Cube/z_error", "Cube/dims_error", "Cube/xy_error", "Cube/conf", "Cube/total_3D_loss", "fast_rcnn/cls_accuracy"):
        assert abs(logs[name] - gold["logs"][name]) <= 2e-4 * max(1.0, abs(gold["logs"][name])), name
    # gradients
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    worst = 0.0
    for n, ref_norm in gold["grad_norm"].items():
        assert n in grads, n
        got = float(grads[n].float().norm())
        rel = abs(got - ref_norm) / max(ref_norm, 1e-6)
        worst = max(worst, rel)
        # fp32 with a different summation order (MFMA k-order, atomics) through ~60 BN layers of a random-init net
        tol = 1e-2 if (n.startswith('roi_heads') or n.startswith('proposal_generator') or 'fpn' in n) else 1e-1
        assert rel < tol or abs(got - ref_norm) < 1e-6, (n, got, ref_norm)
    for n, head in gold["grad_head"].items():
        g = grads[n]
        if g.dim() == 4:   # reference order is (K, C, R, S) row-major
            g = g.contiguous(memory_format=torch.contiguous_format)
        got = g.reshape(g.shape[0], -1).flatten()[:64].cpu() if g.dim() > 1 else g.flatten()[:64].cpu()
        scale = max(head.abs().max().item(), 1e-6)
        tol = 1e-2 if (n.startswith('roi_heads') or n.startswith('proposal_generator') or 'fpn' in n) else 1e-1
        assert (got - head).abs().max().item() <= tol * scale + 1e-7, (n, (got - head).abs().max().item(), scale)
    return worst


@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="~5 min under the host emulator; set OMNI_SLOW=1 (the GPU variant is the gate)")
def test_training_step_matches_reference_emulated(emu_lib):
    _run("cpu", "dla34_tiny")     # 1 image 64x64: the whole step through the host-emulated kernels


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dla34_tiny", "dla34_small", "resnet34_small"])
def test_training_step_matches_reference_gpu(hip_lib, deterministic_forward, name):
    _run("cuda", name)


def _vs_cpu_oracle(batch, report=None):
    """HIP path on the GPU vs the CPU oracle (oracle/model_oracle.py, itself pinned to the reference by
    tests/test_oracle_pin.py) on the same batch, weights and injected sampling variates."""
    from oracle import make_golden as MG
    from oracle import model_oracle as MO
    from omni3d_amd import synthetic
    priors = synthetic.make_priors(50)
    model = MG.build_product_model(MG.product_cfg([]), priors, 5, device="cpu")
    oracle = MO.ModelOracle(priors)
    oracle.load_state_dict(model.state_dict(), strict=True)
    model = model.to("cuda")
    B = len(batch)
    Hp = -(-max(b["image"].shape[1] for b in batch) // 64) * 64      # ImageList pads to the FPN size divisibility
    Wp = -(-max(b["image"].shape[2] for b in batch) // 64) * 64
    A = 3 * sum((Hp // s) * (Wp // s) for s in (4, 8, 16, 32, 64))
    g = torch.Generator().manual_seed(3)
    E_rpn, E_roi = torch.empty(B, A).exponential_(generator=g), torch.empty(B, 2048).exponential_(generator=g)
    model.train()
    oracle.train()
    ref = oracle(batch, E_rpn, E_roi)
    sum(ref.values()).backward()
    # Stage-wise comparison.  Proposal scores of a random-init RPN are nearly tied, so a last-bit difference in one logit
    # (any change of fp32 summation order) can swap two proposals; as the sampling variates are dealt per proposal INDEX,
    # one swap re-deals every later draw and dozens of (equally valid) background ROIs change.  So: (1) the first stage's
    # own proposal list must agree with the oracle's up to a few such swaps, (2) the second stage runs on the oracle's
    # list and must then reproduce the oracle's sampled ROI set exactly and all ten losses to 3e-4.
    model.proposal_generator.injected = {"E": E_rpn, "proposals": oracle.last_proposals}
    model.roi_heads.injected = {"E": E_roi}
    losses = model(batch)
    sum(losses.values()).backward()
    assert torch.equal(model.proposal_generator.last_labels.cpu(), oracle.last_labels)
    own, cnt = model.proposal_generator.last["boxes"].cpu(), model.proposal_generator.last["count"].tolist()
    for n, want in enumerate(oracle.last_proposals):
        got = own[n, :cnt[n]]
        assert abs(len(got) - len(want)) <= 2, (len(got), len(want))
        d = (got[:, None, :] - want[None, :, :]).abs().amax(dim=2)
        assert int((d.min(dim=0).values > 1e-2).sum()) <= 0.01 * len(want) + 2      # membership
        assert int((d.min(dim=1).values > 1e-2).sum()) <= 0.01 * len(got) + 2       # (both directions; clipped proposals of the
                                                                                      # padded area coincide, so ranks are not compared)
    for got, cls, want in zip(model.roi_heads.last_sampled_boxes.cpu(), model.roi_heads.last_sampled_classes.cpu(), oracle.last_roi_boxes):
        got = got[cls >= 0]                                  # unused slots of the fixed 512-block carry class < 0
        assert len(got) == len(want)
        d = (got[:, None, :] - want[None, :, :]).abs().amax(dim=2)
        assert float(d.min(dim=1).values.max()) <= 1e-3 and float(d.min(dim=0).values.max()) <= 1e-3     # same ROI set
    for k, v in ref.items():
        assert abs(float(losses[k].detach()) - float(v.detach())) <= 3e-4 * max(1.0, abs(float(v))), (k, float(losses[k]), float(v))
    og = dict(oracle.named_parameters())
    rows = []
    for n, p in model.named_parameters():
        if p.grad is None or og[n].grad is None:
            continue
        a, b = float(p.grad.float().norm()), float(og[n].grad.norm())
        g = p.grad.float().cpu()
        if g.dim() == 4:
            g = g.contiguous(memory_format=torch.contiguous_format)
        cos = float(torch.nn.functional.cosine_similarity(g.flatten(), og[n].grad.flatten(), dim=0))
        rows.append((abs(a - b) / max(b, 1e-12), cos, n, a, b))
    out = os.path.join(ROOT, "gpurun_out")
    if report and os.path.isdir(out):
        with open(os.path.join(out, report), "w") as f:
            for r in sorted(rows, reverse=True):
                f.write("%.3e cos=%.6f %s %.6g %.6g\n" % r)
    # The gradient of a random-init 60-layer BN network is ill-conditioned towards the stem (ReLU / max-pool /
    # chamfer-argmin decisions flip under 1e-7 perturbations), so the bar tightens with depth: heads 2 %,
    # everything 10 % in norm and direction (cosine) -- the losses above are the fp32 1e-4-class check.
    for rel, cos, n, a, b in rows:
        tight = n.startswith("roi_heads") or n.startswith("proposal_generator") or "fpn" in n
        assert rel <= (2e-2 if tight else 1e-1) or abs(a - b) < 1e-6, (n, a, b)
        if b > 1e-6:
            assert cos > (0.999 if tight else 0.98), (n, cos)


@pytest.mark.gpu
def test_training_step_fullsize_vs_cpu_oracle(hip_lib, deterministic_forward):
    """BASELINE configs[1] exactly: 4 synthetic 512x512 images, default cubercnn_DLA34_FPN config (65 472 anchors, 2000/1000
    proposals, 512 ROIs/img) -- the benchmarked shape, so every kernel choice the bench makes (Winograd F(4x4,3x3) on p2/p3,
    DLA level 2/3, stem kernels, split-K, persistent GEMM) is the one compared with the CPU oracle here."""
    from omni3d_amd import synthetic
    priors = synthetic.make_priors(50)
    _vs_cpu_oracle(synthetic.make_batch(4, 512, 512, num_gt=8, seed=21, priors=priors), report="fullsize_grad_report.txt")


@pytest.mark.gpu
def test_training_step_ragged_batch_vs_cpu_oracle(hip_lib, deterministic_forward):
    """Ragged input: images of different sizes and GT counts in one batch (ImageList zero-pads to the per-batch maximum
    rounded up to 64; anchors cover the padding, proposals are clipped to each image's own size)."""
    from omni3d_amd import synthetic
    priors = synthetic.make_priors(50)
    batch = (synthetic.make_batch(1, 128, 192, num_gt=3, seed=31, priors=priors)
             + synthetic.make_batch(1, 160, 100, num_gt=6, seed=32, priors=priors))
    _vs_cpu_oracle(batch)


@pytest.mark.gpu
def test_training_step_image_without_valid_gt(hip_lib):
    """An image whose only annotation is an ignore region (the reference's Matcher indexing fails on it, SURVEY.md A.8;
    real training filters such images): the HIP path must stay finite -- no foreground, RPN/box losses from background only."""
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    priors = synthetic.make_priors(50)
    model = MG.build_product_model(MG.product_cfg([]), priors, 5, device="cuda")
    batch = synthetic.make_batch(1, 128, 128, num_gt=4, seed=41, priors=priors) + \
        synthetic.make_batch(1, 128, 128, num_gt=1, num_ignore=1, seed=42, priors=priors)
    model.train()
    losses = model(batch)
    total = sum(losses.values())
    total.backward()
    assert all(bool(torch.isfinite(v)) for v in losses.values()), {k: float(v) for k, v in losses.items()}
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
