"""Evaluator core: greedy detection <-> ground-truth matching of `Omni3Deval.evaluateImg`
(/root/reference/cubercnn/evaluation/omni3d_evaluation.py:1433-1551).  tests/golden/eval_match.pt holds outputs of the
REFERENCE method itself (oracle/make_golden.py --eval); the numpy oracle is pinned to it, the HIP kernel to both.
Index work: everything is compared exactly."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT


def _gold():
    return torch.load(os.path.join(ROOT, "tests", "golden", "eval_match.pt"), weights_only=False)


def _ref_to_indices(out, G, D, T):
    """reference ids -> indices: gt ids are 1000 + original index, dt ids 5000 + index, 0 = unmatched"""
    gtind = np.asarray(out["gtIds"], dtype=np.int64) - 1000 if G else np.zeros(0, dtype=np.int64)
    dtm = np.asarray(out["dtMatches"]).astype(np.int64).reshape(T, D)
    gtm_sorted = np.asarray(out["gtMatches"]).astype(np.int64).reshape(T, G)
    dt_match = np.where(dtm > 0, dtm - 1000, -1)
    gt_match = -np.ones_like(gtm_sorted)
    if G:
        gt_match[:, gtind] = np.where(gtm_sorted > 0, gtm_sorted - 5000, -1)
    return gtind, np.asarray(out["gtIgnore"]).astype(np.int64), dt_match, gt_match, np.asarray(out["dtIgnore"]).astype(bool).reshape(T, D)


def test_oracle_matches_reference_evaluateImg():
    from oracle import eval_oracle as EO
    gold = _gold()
    n = 0
    for case in gold["cases"]:
        D, G = len(case["dt_range"]), len(case["gt_range"])
        for a, out in zip(gold["area_rngs"], case["out"]):
            if out is None:
                assert D == 0 and G == 0
                continue
            got = EO.evaluate_img(case["ious"], case["gt_ignore"], case["gt_range"], case["dt_range"], a, gold["iou_thrs"])
            gtind, gtIg, dt_match, gt_match, dtIg = _ref_to_indices(out, G, D, len(gold["iou_thrs"]))
            assert np.array_equal(got["gtind"], gtind) and np.array_equal(got["gtIg"], gtIg)
            assert np.array_equal(got["dt_match"], dt_match) and np.array_equal(got["gt_match"], gt_match)
            assert np.array_equal(got["dtIg"], dtIg)
            n += 1
    assert n > 100


def _kernel_vs(dev, use_gold=True, seed=0, groups=25):
    from oracle import eval_oracle as EO
    from omni3d_amd.cubercnn.evaluation.omni3d_evaluation import evaluate_groups
    gold = _gold()
    thrs, areas = gold["iou_thrs"], gold["area_rngs"]
    if use_gold:
        cases = gold["cases"]
    else:
        rs = np.random.RandomState(seed)
        cases = []
        for _ in range(groups):
            D, G = int(rs.randint(0, 101)), int(rs.randint(0, 80))
            ious = rs.uniform(0, 1, size=(D, G)).astype(np.float32)
            ious[rs.uniform(size=(D, G)) < 0.5] = 0.0
            ious = np.round(ious * 8) / 8 if rs.uniform() < 0.5 else ious          # many exact ties
            cases.append({"ious": ious, "gt_ignore": (rs.uniform(size=G) < 0.3).astype(np.int64), "gt_range": rs.uniform(1, 60, size=G),
                          "dt_range": rs.uniform(1, 60, size=D)})
    dt_sizes = [len(c["dt_range"]) for c in cases]
    gt_sizes = [len(c["gt_range"]) for c in cases]
    flat = torch.from_numpy(np.concatenate([c["ious"].astype(np.float32).reshape(-1) for c in cases] + [np.zeros(0, np.float32)])).to(dev)
    cat = lambda k, dt: torch.from_numpy(np.concatenate([np.asarray(c[k], dtype=dt) for c in cases] + [np.zeros(0, dt)])).to(dev)  # noqa: E731
    res = evaluate_groups(flat, dt_sizes, gt_sizes, cat("gt_ignore", np.int32), cat("gt_range", np.float32), cat("dt_range", np.float32),
                          areas, thrs)
    do, go = 0, 0
    for gi, c in enumerate(cases):
        D, G = dt_sizes[gi], gt_sizes[gi]
        for ai, a in enumerate(areas):
            # the kernel sees fp32 ranges: give the oracle the same values
            want = EO.evaluate_img(c["ious"], c["gt_ignore"], np.asarray(c["gt_range"], np.float32), np.asarray(c["dt_range"], np.float32),
                                   a, thrs)
            assert np.array_equal(res["dt_match"][ai, :, do:do + D].cpu().numpy(), want["dt_match"]), (gi, ai)
            assert np.array_equal(res["gt_match"][ai, :, go:go + G].cpu().numpy(), want["gt_match"]), (gi, ai)
            assert np.array_equal(res["dt_ignore"][ai, :, do:do + D].cpu().numpy().astype(bool), want["dtIg"]), (gi, ai)
            assert np.array_equal(res["gt_order"][ai, go:go + G].cpu().numpy(), want["gtind"]), (gi, ai)
            assert np.array_equal(res["gt_ignore"][ai, go:go + G].cpu().numpy()[want["gtind"]], want["gtIg"]), (gi, ai)
        do, go = do + D, go + G


def test_eval_match_emulated(emu_lib):
    _kernel_vs("cpu", use_gold=True)
    _kernel_vs("cpu", use_gold=False, seed=3, groups=6)


@pytest.mark.gpu
def test_eval_match_gpu(hip_lib):
    _kernel_vs("cuda", use_gold=True)
    _kernel_vs("cuda", use_gold=False, seed=4, groups=200)
