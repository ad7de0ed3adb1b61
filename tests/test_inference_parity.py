"""Eval-mode parity (RCNN3D.inference, /root/reference/cubercnn/modeling/meta_arch/rcnn3d.py:79-112;
fast_rcnn_inference_single_image fast_rcnn.py:57-116; cube eval branch roi_heads.py:353-357,771-824) against
the fixture produced by the reference's own files (oracle/make_golden.py --infer)."""
import os

import pytest
import torch

from conftest import ROOT

# default head; Z_TYPE clusters + CLUSTER_BINS 4 + SCALE_ROI_BOXES; ground-truth 2D boxes handed in as `oracle2D` (no RPN / box head)
FIXTURES = ["dla34_small_infer", "dla34_small_infer_clusters", "dla34_small_infer_oracle2d"]
# the BENCHMARKED inference shape (bench.py --workload infer): 4 x 512 x 512, 1000 proposals / image, the configuration's own test
# settings.  At this size the reference's own fp32 run and its fp64 run agree on only 81 .. 94 of the 100 detections per image
# (near-ties at the NMS threshold and at the top-100 cut between 50 000 (proposal, class) candidates of a random-init classifier),
# so "the same detection list" is not defined to better than that: the fixture records which fp32 detections have an fp64 twin,
# the test requires the HIP list to share at least as many detections with the fp32 reference (minus 10) and bounds every
# value on the detections all three runs have in common.
FULL_FIXTURES = ["dla34_full_infer"]


def _run(dev, name="dla34_small_infer"):
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    gold = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
    spec = gold["spec"]
    priors = synthetic.make_priors(50, bins=spec.get("prior_bins", 0))
    model = MG.sharpen(MG.build_product_model(MG.product_cfg(spec["overrides"]), priors, spec["seed"])).to(dev)
    batch = MG.infer_batch(spec, priors)
    model.eval()
    with torch.no_grad():
        out = model(batch)
    assert len(out) == len(gold["results"])
    report, bad = [], []
    loose = bool(spec.get("selection_may_differ"))
    for o, ref in zip(out, gold["results"]):
        i = o["instances"]
        n = len(ref["scores"])
        assert len(i) == n, (len(i), n)
        r64 = ref["fp64"]                                                               # the reference files in float64
        if not loose:
            assert torch.equal(i.pred_classes.cpu().long(), ref["pred_classes"])        # selection is index-exact
        else:
            # twin of every reference detection in the HIP list: same class, same box to half a pixel
            hb, hc = i.pred_boxes.tensor.cpu(), i.pred_classes.cpu().long()
            d = (ref["pred_boxes"][:, None] - hb[None]).abs().amax(2) + 1e6 * (ref["pred_classes"][:, None] != hc[None])
            m, found = d.argmin(1), d.min(1).values < 0.5
            own = int(ref["fp64_found"].sum())
            report.append("detections shared with the fp32 reference: %d of %d (the reference's own fp32 / fp64 runs share %d)" % (int(found.sum()), n, own))
            assert int(found.sum()) >= own - 10, report[-1]
            both = found & ref["fp64_found"]
            sel = m[both]
            # compare on the common detections: gather the HIP rows, restrict the reference rows
            i = i[sel.to(i.pred_classes.device)]
            ref = {k: (v[both] if torch.is_tensor(v) and v.shape[:1] == both.shape else v) for k, v in ref.items()}
            r64 = {k: v[both] for k, v in r64.items()}

        def bounded(name, got, cap, scale=None):
            """|HIP - fp64| <= max(cap, 3 x the reference's own fp32 distance to the fp64 value), never above max(2 x cap, 1.5 x that
            distance) (cap = north_star's 1e-4, relative for values above 1; the second term only matters where the REFERENCE's own
            fp32 run is above the bar -- the 6D pose of one detection of the small fixture, 1.9e-4).  The split-K / fc1 atomics reorder the fp32 sums from
            run to run: over repeated runs the worst element of pred_dimensions sits between 2e-5 and 1.1e-4 while the CPU
            fp32 reference is a steady 4.5e-5 from float64 (profiles/r02_parity_fp64_inference.txt) -- the MAX over all
            detections of an fp32 rounding error is what fluctuates, so the bar is tied to the measured conditioning."""
            got, r32, r_64 = got.double().cpu(), ref[name].double(), r64[name].double()
            den = (1.0 + r_64.abs()) if scale is None else scale
            e_hip, e_ref = float(((got - r_64).abs() / den).max()), float(((r32 - r_64).abs() / den).max())
            report.append("%-16s |hip-fp64| %.2e  |ref32-fp64| %.2e  cap %.0e" % (name, e_hip, e_ref, cap))
            if not (e_hip <= max(cap, 3.0 * e_ref) and e_hip <= max(2.0 * cap, 1.5 * e_ref)):
                bad.append(report[-1])

        ext = float(max(o["instances"].image_size))
        bounded("scores", i.scores, 1e-4)
        bounded("pred_boxes", i.pred_boxes.tensor, 1e-4, scale=ext)          # pixel quantities: 1e-4 of the image extent
        bounded("pred_dimensions", i.pred_dimensions, 1e-4)
        bounded("pred_center_cam", i.pred_center_cam, 1e-4)
        # projected 3D centres may lie far outside the image (|u| up to ~800 px here); round 6: north_star's 1e-4 (4e-4 before; measured <= 2.1e-5)
        bounded("pred_center_2D", i.pred_center_2D, 1e-4, scale=r64["pred_center_2D"].abs().clamp(min=ext))
        # Round 4: north_star's 1e-4 here too (1e-3 before).  The Gram-Schmidt of a random-init 6D pose amplifies feature noise by
        # up to ~300x (angle between the two pose vectors 3 degrees: tools/probes/pose_diag.py, profiles/r04_pose_diag.txt); the decode
        # kernel evaluates the fp32 head outputs in float64 (arithmetic error 3e-8) and inference runs its Winograd layers on the
        # 16-point transform, which brought the full-size fixture from 2.9e-4 to 3.5e-5.  Where the REFERENCE's fp32 run is itself
        # above the bar (small fixture, 1.9e-4 / 3.1e-4) the 3x / 1.5x rule of `bounded` applies.
        bounded("pred_pose", i.pred_pose, 1e-4)
        # corners = centre +- R dims / 2 cancel for the random-init head's cuboids of up to ~100 m around a centre a few metres
        # away, so the error of a corner is measured against the extent of ITS cuboid, not against the corner coordinate
        bounded("pred_bbox3D", i.pred_bbox3D, 1e-4, scale=1.0 + r64["pred_bbox3D"].double().abs().amax(dim=(1, 2), keepdim=True))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir) and dev == "cuda":
        with open(os.path.join(out_dir, "inference_fp64_report.txt" if name == "dla34_small_infer" else name + "_fp64_report.txt"), "w") as f:
            f.write("\n".join(report) + "\n")
    assert not bad, "\n".join(bad)


@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="minutes under the host emulator; set OMNI_SLOW=1 (the GPU variant is the gate)")
@pytest.mark.parametrize("name", FIXTURES)
def test_inference_matches_reference_emulated(emu_lib, name):
    _run("cpu", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES + FULL_FIXTURES)
def test_inference_matches_reference_gpu(hip_lib, name):
    _run("cuda", name)


FIELDS_OUT = ("scores", "scores_full", "pred_classes", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose")


def _same_results(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        i, j = x["instances"], y["instances"]
        assert len(i) == len(j) and i.image_size == j.image_size
        assert torch.equal(i.pred_boxes.tensor, j.pred_boxes.tensor)
        for f in FIELDS_OUT:
            assert torch.equal(getattr(i, f), getattr(j, f)), f


@pytest.mark.gpu
def test_replayed_inference_equals_eager_gpu(hip_lib):
    """meta_arch/infer_replay.py: `model(batch)` in eval mode from the captured hipGraph of its size bucket gives the eager pass's
    results bit for bit -- also for a second batch of the bucket with other images, other (ragged) sizes and other intrinsics, which
    must reach the static tensors -- and the batched postprocess equals the per-image form of the reference"""
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.modeling.meta_arch import infer_replay
    from omni3d_amd.cubercnn.modeling.roi_heads import inference as INF
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "dla34_small_infer.pt"), weights_only=False)
    spec = gold["spec"]
    priors = synthetic.make_priors(50, bins=spec.get("prior_bins", 0))
    model = MG.sharpen(MG.build_product_model(MG.product_cfg(spec["overrides"]), priors, spec["seed"])).to("cuda")
    model.eval()
    first = MG.infer_batch(spec, priors)
    H, W = first[0]["image"].shape[-2:]
    second = synthetic.make_batch(len(first), H, W, num_gt=4, seed=77, priors=priors)
    second[0]["image"] = second[0]["image"][:, : H - 24, : W - 40].contiguous()          # ragged, same 64-pixel bucket
    second[0]["height"], second[0]["width"] = H - 24, W - 40
    second[-1]["K"] = [[700.0, 0.0, W / 2.0 + 3.0], [0.0, 700.0, H / 2.0 - 2.0], [0.0, 0.0, 1.0]]
    for b in first + second:
        b["image"] = b["image"].to("cuda")
        b.pop("instances", None)
    prev = infer_replay.ENABLED
    try:
        infer_replay.ENABLED = False
        with torch.no_grad():
            eager = [model(first), model(second)]
        infer_replay.ENABLED = True
        with torch.no_grad():
            model(first)                                  # pass 1 of the bucket: eager, fills the per-shape caches
            replayed = [model(first), model(second), model(first)]
        rep = model.__dict__["_omni_infer"]
        assert rep.failed is None and rep.captures == 1 and rep.replays == 3, (rep.failed, rep.captures, rep.replays)
    finally:
        infer_replay.ENABLED = prev
    _same_results(eager[0], replayed[0])
    _same_results(eager[1], replayed[1])
    _same_results(eager[0], replayed[2])
    # a parameter write drops the captured passes (they hold the BatchNorm coefficients of the old values)
    with torch.no_grad():
        bn = next(m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
        bn.running_var.mul_(1.5)
        moved = model(first)
        assert rep.captures == 1 and len(rep.cache) == 0          # eager again (first pass of the bucket after the write)
        infer_replay.ENABLED = False
        want = model(first)
        infer_replay.ENABLED = True
        again = model(first)
        assert rep.captures == 2
        bn.running_var.div_(1.5)
    _same_results(want, moved)
    _same_results(want, again)
    assert sum(len(r["instances"]) for r in eager[0]) > 0
    # the batched postprocess against the per-image form on the same raw results
    with torch.no_grad():
        res = model.inference(first, do_postprocess=False)
    for r in res:
        assert getattr(r, "_omni_slots", None) is not None
    sizes = [(b["image"].shape[-2], b["image"].shape[-1]) for b in first]
    fast = INF.postprocess(res, first, sizes)
    for r in res:
        r.__dict__.pop("_omni_slots")
    slow = INF.postprocess(res, first, sizes)
    _same_results(fast, slow)


def test_batched_postprocess_equals_per_image_form_cpu():
    """roi_heads/inference.py:_postprocess_slots (one rescale / clip / empty-box pass for the batch) against the per-image form of
    detectron2's detector_postprocess -- including an image whose clipped boxes are partly EMPTY (the boolean-indexing fallback) and an
    image without detections"""
    from omni3d_amd.cubercnn.modeling.roi_heads import inference as INF
    g = torch.Generator().manual_seed(5)
    B, topk, K = 3, 6, 4
    dbox = torch.rand(B, topk, 4, generator=g) * 60.0
    dbox[..., 2:] += dbox[..., :2] + 1.0
    dbox[0, 1] = torch.tensor([70.0, 10.0, 90.0, 30.0])          # left of nothing: clipped to zero width at W = 64 -> dropped
    dbox[0, 3] = torch.tensor([10.0, 80.0, 30.0, 95.0])          # below the image: zero height after the clip -> dropped
    raw = {"dbox": dbox, "final": torch.rand(B * topk, generator=g), "full": torch.rand(B, topk, K, generator=g),
           "dcls": torch.randint(0, K, (B, topk), generator=g, dtype=torch.int32), "verts": torch.rand(B * topk, 8, 3, generator=g),
           "cube3d": torch.rand(B * topk, 9, generator=g), "pose": torch.rand(B * topk, 3, 3, generator=g),
           "dcount": torch.tensor([5, 0, 6], dtype=torch.int32)}
    sizes = [(64, 64), (48, 64), (64, 56)]
    infos = [{"height": 128, "width": 128}, {"height": 48, "width": 64}, {"height": 96, "width": 84}]
    res = INF.collect_detections(raw, sizes)
    assert [len(r) for r in res] == [5, 0, 6] and all(getattr(r, "_omni_slots", None) is not None for r in res)
    fast = INF.postprocess(res, infos, sizes)
    for r in res:
        r.__dict__.pop("_omni_slots")
    slow = INF.postprocess(res, infos, sizes)
    _same_results(fast, slow)
    assert len(fast[0]["instances"]) == 3 and len(fast[1]["instances"]) == 0 and len(fast[2]["instances"]) == 6
    assert fast[2]["instances"].image_size == (96, 84)


def test_replayed_inference_staging_emulated(emu_lib):
    """InferReplay with graphs=False (the same slots / static packed description / digest logic, eager launches): a second batch of
    the bucket with a ragged image and other intrinsics must reach the static tensors, results equal the plain eager pass, and a
    parameter write drops the captured passes"""
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.modeling.meta_arch import infer_replay
    LIGHT = ["MODEL.RPN.PRE_NMS_TOPK_TEST", 60, "MODEL.RPN.POST_NMS_TOPK_TEST", 20, "MODEL.DLA.TYPE", "dla46_c", "MODEL.FPN.OUT_CHANNELS", 32,
             "MODEL.ROI_BOX_HEAD.FC_DIM", 64, "MODEL.ROI_CUBE_HEAD.FC_DIM", 64, "TEST.DETECTIONS_PER_IMAGE", 10]
    priors = synthetic.make_priors(50)
    model = MG.sharpen(MG.build_product_model(MG.product_cfg(LIGHT), priors, 11, device="cpu"))
    model.eval()
    first = synthetic.make_batch(2, 64, 64, num_gt=3, seed=5, priors=priors)
    second = synthetic.make_batch(2, 64, 64, num_gt=3, seed=6, priors=priors)
    second[1]["image"] = second[1]["image"][:, :40, :56].contiguous()
    second[1]["height"], second[1]["width"] = 80, 112
    second[0]["K"] = [[300.0, 0.0, 30.0], [0.0, 300.0, 34.0], [0.0, 0.0, 1.0]]
    for b in first + second:
        b.pop("instances", None)
    prev = infer_replay.ENABLED
    try:
        infer_replay.ENABLED = False
        with torch.no_grad():
            eager = [model(first), model(second)]
        infer_replay.ENABLED = True
        rep = model.__dict__["_omni_infer"] = infer_replay.InferReplay(model, graphs=False)
        with torch.no_grad():
            model(first)
            got = [model(first), model(second), model(first)]
            assert rep.failed is None and rep.captures == 1 and rep.replays == 3, (rep.failed, rep.captures, rep.replays)
            _same_results(eager[0], got[0])
            _same_results(eager[1], got[1])
            _same_results(eager[0], got[2])
            assert sum(len(r["instances"]) for r in eager[0] + eager[1]) > 0
            next(m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)).running_var.mul_(1.25)
            model(first)
            assert len(rep.cache) == 0 and rep.captures == 1
    finally:
        infer_replay.ENABLED = prev
