"""oracle/eval_oracle.py -- TEST INFRASTRUCTURE.  numpy restatement of the greedy detection <-> ground-truth matching of
`Omni3Deval.evaluateImg` (/root/reference/cubercnn/evaluation/omni3d_evaluation.py:1433-1551; 3D mode, eval_prox off)
for ONE (image, category) group and ONE area (depth) range.  Pinned to the reference function itself by
tests/golden/eval_match.pt (oracle/make_golden.py --eval runs the reference method on the same inputs)."""
import numpy as np


def evaluate_img(ious, gt_ignore, gt_range, dt_range, a_rng, iou_thrs):
    """ious (D, G): rows = detections in descending-score order (already cut to maxDet), columns = ground truths in their
    original order.  gt_ignore (G,) 0/1 = `ignore3D`; gt_range (G,), dt_range (D,) = `depth`; a_rng = (lo, hi).
    -> dict: gtind (G,) the stable ignore-last permutation (:1463); gtIg (G,) `_ignore` in that order (:1486);
       dt_match (T, D) / gt_match (T, G): index of the matched gt (ORIGINAL order) / dt, -1 = none (the reference stores
       ids, :1519-1520); dtIg (T, D) bool (:1518, :1527)."""
    ious = np.asarray(ious, dtype=np.float64).reshape(len(dt_range), len(gt_range))
    D, G, T = len(dt_range), len(gt_range), len(iou_thrs)
    ig = np.array([1 if (gt_ignore[g] or gt_range[g] < a_rng[0] or gt_range[g] > a_rng[1]) else 0 for g in range(G)], dtype=np.int64)  # :1455-1459
    gtind = np.argsort(ig, kind="mergesort")                                                   # :1463
    gtIg = ig[gtind]
    io = ious[:, gtind] if G > 0 and D > 0 else ious                                           # :1469-1473
    gtm = -np.ones((T, G), dtype=np.int64)       # sorted order, holds dt index
    dtm = -np.ones((T, D), dtype=np.int64)       # holds sorted gt position
    dtIg = np.zeros((T, D), dtype=bool)
    if D > 0 and G > 0:                                                                        # :1490
        for tind, t in enumerate(iou_thrs):
            for dind in range(D):
                iou = min([t, 1 - 1e-10])                                                      # :1495
                m = -1
                for gind in range(G):
                    if gtm[tind, gind] >= 0:                                                   # :1504 (ids are positive)
                        continue
                    if m > -1 and gtIg[m] == 0 and gtIg[gind] == 1:                            # :1508
                        break
                    if io[dind, gind] < iou:                                                   # :1512
                        continue
                    iou = io[dind, gind]                                                       # :1516-1517
                    m = gind
                if m == -1:
                    continue
                dtIg[tind, dind] = bool(gtIg[m])                                               # :1523-1525
                dtm[tind, dind] = m
                gtm[tind, m] = dind
    a = np.array([dt_range[d] < a_rng[0] or dt_range[d] > a_rng[1] for d in range(D)], dtype=bool).reshape(1, D)   # :1528-1530
    dtIg = np.logical_or(dtIg, np.logical_and(dtm < 0, np.repeat(a, T, 0)))                    # :1532
    dt_match = np.where(dtm >= 0, gtind[np.clip(dtm, 0, max(G - 1, 0))] if G > 0 else -1, -1)
    gt_match = -np.ones((T, G), dtype=np.int64)
    if G > 0:
        gt_match[:, gtind] = gtm
    return {"gtind": gtind, "gtIg": gtIg, "dt_match": dt_match, "gt_match": gt_match, "dtIg": dtIg}
