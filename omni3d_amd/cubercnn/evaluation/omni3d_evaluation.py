"""`box3d_overlap` (reference cubercnn/evaluation/omni3d_evaluation.py:106-166) on the IoU3D kernel, and the batched
form the evaluator needs: `Omni3Deval.evaluate` calls `computeIoU` once per (image, category) group from a Python dict
comprehension (:1339-1343, :1357-1431), i.e. thousands of tiny `box3d_overlap` calls; `box3d_overlap_groups` takes all
groups at once -- one validity launch + one ragged pairs launch + one readback.  `Omni3Deval` runs the COCO-style greedy
matching and the precision / recall accumulation around it on the device as well; `Omni3DEvaluator` / `Omni3DEvaluationHelper`
are the per-split drivers `tools/train_net.py:do_test` uses."""
import copy
import datetime
import json
import logging
import os

import numpy as np
import torch

from ...kernels import iou3d


def box3d_overlap(boxes_dt: torch.Tensor, boxes_gt: torch.Tensor, eps_coplanar: float = 1e-4, eps_nonzero: float = 1e-8) -> torch.Tensor:
    """(N,8,3), (M,8,3) corner lists -> (N,M) IoU; rows of non-coplanar / zero-area detections are 0."""
    dev = boxes_dt.device
    if not boxes_dt.is_cuda:   # the reference forces this path to the CPU (MAX_DTS_CROSS_GTS_FOR_IOU3D = 0); here it is a GPU op
        dev = torch.device("cuda")
    out = iou3d.box3d_overlap(boxes_dt.to(dev).float(), boxes_gt.to(dev).float(), eps_coplanar, eps_nonzero)
    return out.to(boxes_dt.device)


def box3d_overlap_groups(boxes_dt, boxes_gt, dt_sizes, gt_sizes, eps_coplanar: float = 1e-4, eps_nonzero: float = 1e-8, warn=False):
    """All (image, category) groups of an evaluation in one pass.

    boxes_dt (sum(dt_sizes), 8, 3) / boxes_gt (sum(gt_sizes), 8, 3): the groups' detection (already score-sorted and cut to
    maxDets, :1374-1377) and ground-truth corner lists, concatenated in group order; dt_sizes / gt_sizes: per-group counts.
    -> list of (Nd_g, Ng_g) float32 IoU matrices (views of one flat device tensor, in group order; empty groups give
    empty matrices), each equal to `box3d_overlap(dt_g, gt_g)` of the reference."""
    dt_sizes = np.asarray(dt_sizes, dtype=np.int64)
    gt_sizes = np.asarray(gt_sizes, dtype=np.int64)
    if dt_sizes.shape != gt_sizes.shape or dt_sizes.ndim != 1:
        raise ValueError("dt_sizes / gt_sizes must be 1-D and of equal length")
    if int(dt_sizes.sum()) != boxes_dt.shape[0] or int(gt_sizes.sum()) != boxes_gt.shape[0]:
        raise ValueError("group sizes do not add up to the number of boxes")
    dev = boxes_dt.device if boxes_dt.is_cuda else torch.device("cuda")
    if not boxes_dt.is_cuda and iou3d._lib.get().emulated:      # host-emulated kernels in the GPU-less test suite
        dev = boxes_dt.device
    dt, gt = boxes_dt.to(dev).float().contiguous(), boxes_gt.to(dev).float().contiguous()
    counts = dt_sizes * gt_sizes
    pair_off = np.concatenate([[0], np.cumsum(counts)])
    P = int(pair_off[-1])
    # ragged pair list, row-major inside each group (host index arithmetic on the group table, one upload)
    gid = np.repeat(np.arange(len(counts)), counts)
    local = np.arange(P) - pair_off[gid]
    ng = np.maximum(gt_sizes[gid], 1)
    dt_off = np.concatenate([[0], np.cumsum(dt_sizes)])[:-1]
    gt_off = np.concatenate([[0], np.cumsum(gt_sizes)])[:-1]
    idx1 = torch.from_numpy((dt_off[gid] + local // ng).astype(np.int32)).to(dev)
    idx2 = torch.from_numpy((gt_off[gid] + local % ng).astype(np.int32)).to(dev)
    if P == 0:
        flat = torch.zeros(0, dtype=torch.float32, device=dev)
    else:
        valid, vcounts = iou3d.box3d_validity(dt, eps_coplanar, eps_nonzero)
        _, flat = iou3d.iou_box3d_pairs(dt, gt, idx1, idx2, valid1=valid)
        if warn:
            c = vcounts.tolist()
            if c[0] > 0:
                print('Warning: skipping {:d} non-coplanar boxes at eval.'.format(int(c[0])))
            if c[1] > 0:
                print('Warning: skipping {:d} zero volume boxes at eval.'.format(int(c[1])))
    return [flat[pair_off[g]:pair_off[g + 1]].view(int(dt_sizes[g]), int(gt_sizes[g])) for g in range(len(counts))]


def evaluate_groups(ious_flat, dt_sizes, gt_sizes, gt_ignore, gt_range, dt_range, area_ranges, iou_thrs):
    """The greedy matching of `Omni3Deval.evaluateImg` (:1433-1551, 3D mode, eval_prox off) for every (image, category)
    group x depth range x IoU threshold in one launch (the reference loops over them in Python, :1346-1351).

    ious_flat: the concatenated (D_g, G_g) matrices (e.g. torch.cat of box3d_overlap_groups' views), detections in descending
    score order and cut to maxDets; dt_sizes / gt_sizes: per-group counts; gt_ignore (sumG) int `ignore3D`; gt_range (sumG),
    dt_range (sumD) float `depth`; area_ranges (A, 2); iou_thrs (T,).
    -> dict of device tensors: dt_match (A,T,sumD) index of the matched gt inside its group (original order) or -1,
       gt_match (A,T,sumG) index of the matched dt or -1, dt_ignore (A,T,sumD) uint8, gt_order (A,sumG) the stable
       ignore-last order, gt_ignore (A,sumG) uint8 `_ignore` per original gt."""
    L = iou3d._lib.get()
    dev = ious_flat.device
    dt_sizes, gt_sizes = np.asarray(dt_sizes, dtype=np.int64), np.asarray(gt_sizes, dtype=np.int64)
    ng, sumD, sumG = len(dt_sizes), int(dt_sizes.sum()), int(gt_sizes.sum())
    A, T = len(area_ranges), len(iou_thrs)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)      # noqa: E731
    iou_off = torch.from_numpy(np.concatenate([[0], np.cumsum(dt_sizes * gt_sizes)])[:-1].astype(np.int64)).to(dev)
    dt_off, gt_off = i32(np.concatenate([[0], np.cumsum(dt_sizes)])), i32(np.concatenate([[0], np.cumsum(gt_sizes)]))
    areas = torch.tensor(np.asarray(area_ranges, dtype=np.float32)).to(dev).contiguous()
    thrs = torch.tensor(np.asarray(iou_thrs, dtype=np.float64)).to(dev).contiguous()
    out = {"dt_match": torch.empty((A, T, sumD), dtype=torch.int32, device=dev), "gt_match": torch.empty((A, T, sumG), dtype=torch.int32, device=dev),
           "dt_ignore": torch.empty((A, T, sumD), dtype=torch.uint8, device=dev), "gt_order": torch.empty((A, sumG), dtype=torch.int32, device=dev),
           "gt_ignore": torch.empty((A, sumG), dtype=torch.uint8, device=dev)}
    if ng == 0:
        return out
    ious_flat = ious_flat.float().contiguous()
    gt_ignore, gt_range, dt_range = gt_ignore.to(torch.int32).contiguous(), gt_range.float().contiguous(), dt_range.float().contiguous()
    iou3d._lib.check_device(ious_flat, gt_ignore, gt_range, dt_range)
    _p = iou3d._lib.ptr
    L.call("omni_eval_match", _p(ious_flat), _p(iou_off), _p(dt_off), _p(gt_off), _p(gt_ignore), _p(gt_range), _p(dt_range), _p(areas),
           _p(thrs), ng, A, T, sumD, sumG, int(gt_sizes.max()) if ng else 0, _p(out["dt_match"]), _p(out["gt_match"]),
           _p(out["dt_ignore"]), _p(out["gt_order"]), _p(out["gt_ignore"]), iou3d._lib.stream_of(ious_flat))
    return out


# =====================================================================================================================
# Omni3Deval: evaluate -> accumulate -> summarize on the device (reference :1019-1704)
# =====================================================================================================================
class Omni3DParams:
    """omni3d_evaluation.py:1019-1088"""

    def setDet2DParams(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.recThrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ["all", "small", "medium", "large"]
        self.useCats = 1

    def setDet3DParams(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(0.05, 0.5, int(np.round((0.5 - 0.05) / 0.05)) + 1, endpoint=True)
        self.recThrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0, 1e5], [0, 10], [10, 35], [35, 1e5]]
        self.areaRngLbl = ["all", "near", "medium", "far"]
        self.useCats = 1

    def __init__(self, mode="2D"):
        if mode == "2D":
            self.setDet2DParams()
        elif mode == "3D":
            self.setDet3DParams()
        else:
            raise Exception("mode %s not supported" % (mode))
        self.iouType = "bbox"
        self.mode = mode
        self.proximity_thresh = 0.3


class AnnotationIndex:
    """The four COCO-API calls Omni3Deval makes (getImgIds / getCatIds / getAnnIds / loadAnns) over a plain list of
    annotation dicts -- pycocotools is not needed for the scoped path."""

    def __init__(self, anns, img_ids=None, cat_ids=None):
        self.anns = {}
        for i, a in enumerate(anns):
            a.setdefault("id", i + 1)
            self.anns[a["id"]] = a
        self._imgs = sorted(set(img_ids) if img_ids is not None else {a["image_id"] for a in anns})
        self._cats = sorted(set(cat_ids) if cat_ids is not None else {a["category_id"] for a in anns})

    def getImgIds(self):
        return list(self._imgs)

    def getCatIds(self):
        return list(self._cats)

    def getAnnIds(self, imgIds=(), catIds=()):
        imgs, cats = set(imgIds), set(catIds)
        return [i for i, a in self.anns.items() if (not imgs or a["image_id"] in imgs) and (not cats or a["category_id"] in cats)]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]


def _ragged_pairs(dt_sizes, gt_sizes):
    counts = dt_sizes * gt_sizes
    pair_off = np.concatenate([[0], np.cumsum(counts)])
    P = int(pair_off[-1])
    gid = np.repeat(np.arange(len(counts)), counts)
    local = np.arange(P) - pair_off[gid]
    ng = np.maximum(gt_sizes[gid], 1)
    dt_off = np.concatenate([[0], np.cumsum(dt_sizes)])[:-1]
    gt_off = np.concatenate([[0], np.cumsum(gt_sizes)])[:-1]
    return (dt_off[gid] + local // ng).astype(np.int64), (gt_off[gid] + local % ng).astype(np.int64), pair_off


class Omni3Deval:
    """`Omni3Deval(cocoGt, cocoDt, mode=...)` with the reference's evaluate() / accumulate() / summarize() and result layout
    (`eval['precision']` [T,R,K,A,M], `eval['recall']` [T,K,A,M], `eval['scores']`, `stats` (13,)).

    evaluate(): all (image, category) groups at once -- one IoU pass (`box3d_overlap_groups` in 3D, a vectorised box IoU in
    2D) and one greedy-matching launch for every group x range x threshold (`evaluate_groups`, csrc/eval_match.hip);
    accumulate(): one launch for every (category, range, maxDets, threshold) (`omni_eval_accumulate`).  The reference runs
    these as Python loops over dict-of-list structures (:1339-1351, :1230-1301).

    eval_prox (proximity evaluation for non-exhaustively annotated datasets, :1419-1429, :1499, :1534-1536): a detection may
    only match ground truths whose 2D box overlaps its own by more than `params.proximity_thresh`, and a detection with no
    ground truth in proximity is ignored.  True / False like the reference, or a collection of image ids (extension: the
    helper evaluates the union of several datasets of which only some use proximity evaluation)."""

    def __init__(self, cocoGt=None, cocoDt=None, iouType="bbox", mode="2D", eval_prox=False):
        if mode not in ["2D", "3D"]:
            raise Exception("mode %s not supported" % (mode))
        self.mode, self.eval_prox = mode, eval_prox
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.params = Omni3DParams(mode)
        self.eval, self.stats, self._dev = {}, [], None
        if cocoGt is not None:
            self.params.imgIds = sorted(cocoGt.getImgIds())
            self.params.catIds = sorted(cocoGt.getCatIds())

    # ---- evaluate (:1315-1357 + computeIoU :1359-1431 + evaluateImg :1433-1551) ----------------------------------------
    def evaluate(self, device=None):
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        p.catIds = list(np.unique(p.catIds)) if p.useCats else p.catIds
        p.maxDets = sorted(p.maxDets)
        if not p.useCats:
            raise NotImplementedError("useCats = 0 is not built on the device path")
        gts = self.cocoGt.loadAnns(self.cocoGt.getAnnIds(imgIds=p.imgIds, catIds=p.catIds))
        dts = self.cocoDt.loadAnns(self.cocoDt.getAnnIds(imgIds=p.imgIds, catIds=p.catIds))
        flag = "ignore2D" if self.mode == "2D" else "ignore3D"
        g_by, d_by = {}, {}
        for g in gts:
            g[flag] = g[flag] if flag in g else 0
            g_by.setdefault((g["image_id"], g["category_id"]), []).append(g)
        for d in dts:
            d_by.setdefault((d["image_id"], d["category_id"]), []).append(d)
        maxDet = p.maxDets[-1]
        # group table in (category, image) order = the order evalImgs / accumulate walk (:1346-1351, :1230-1241)
        groups = []
        for ki, cat in enumerate(p.catIds):
            for ii, img in enumerate(p.imgIds):
                g, d = g_by.get((img, cat), []), d_by.get((img, cat), [])
                if not g and not d:
                    continue
                order = np.argsort([-x["score"] for x in d], kind="mergesort")
                groups.append((ki, ii, g, [d[i] for i in order[:maxDet]]))
        key = "bbox" if self.mode == "2D" else "bbox3D"
        rng_key = "area" if self.mode == "2D" else "depth"
        dt_sizes = np.array([len(gr[3]) for gr in groups], dtype=np.int64)
        gt_sizes = np.array([len(gr[2]) for gr in groups], dtype=np.int64)
        all_d = [x for gr in groups for x in gr[3]]
        all_g = [x for gr in groups for x in gr[2]]
        if device is None:
            device = torch.device("cpu") if iou3d._lib.get().emulated else torch.device("cuda")
        f32 = lambda v, shape: torch.tensor(np.asarray(v, dtype=np.float32).reshape(shape)).to(device)      # noqa: E731
        if self.mode == "3D":
            mats = box3d_overlap_groups(f32([x[key] for x in all_d], (-1, 8, 3)), f32([x[key] for x in all_g], (-1, 8, 3)), dt_sizes, gt_sizes)
            flat = torch.cat([m.reshape(-1) for m in mats]) if mats else torch.zeros(0, device=device)
        else:
            i1, i2, _ = _ragged_pairs(dt_sizes, gt_sizes)
            bd, bg = f32([x[key] for x in all_d], (-1, 4))[torch.from_numpy(i1).to(device)], f32([x[key] for x in all_g], (-1, 4))[torch.from_numpy(i2).to(device)]
            iw = (torch.min(bd[:, 0] + bd[:, 2], bg[:, 0] + bg[:, 2]) - torch.max(bd[:, 0], bg[:, 0])).clamp(min=0)
            ih = (torch.min(bd[:, 1] + bd[:, 3], bg[:, 1] + bg[:, 3]) - torch.max(bd[:, 1], bg[:, 1])).clamp(min=0)
            inter = iw * ih
            flat = inter / (bd[:, 2] * bd[:, 3] + bg[:, 2] * bg[:, 3] - inter)                  # pycocotools bbIou, iscrowd = 0
        far = None
        if self.eval_prox is not False and self.eval_prox is not None and len(all_d) and len(all_g):
            i1, i2, _ = _ragged_pairs(dt_sizes, gt_sizes)
            t1, t2 = torch.from_numpy(i1).to(device), torch.from_numpy(i2).to(device)
            if self.mode == "2D":
                iou2d = flat
            else:
                bd, bg = f32([x["bbox"] for x in all_d], (-1, 4))[t1], f32([x["bbox"] for x in all_g], (-1, 4))[t2]
                iw = (torch.min(bd[:, 0] + bd[:, 2], bg[:, 0] + bg[:, 2]) - torch.max(bd[:, 0], bg[:, 0])).clamp(min=0)
                ih = (torch.min(bd[:, 1] + bd[:, 3], bg[:, 1] + bg[:, 3]) - torch.max(bd[:, 1], bg[:, 1])).clamp(min=0)
                iou2d = iw * ih / (bd[:, 2] * bd[:, 3] + bg[:, 2] * bg[:, 3] - iw * ih)
            prox = iou2d > p.proximity_thresh
            if self.eval_prox is not True:            # only the images named
                chosen = set(self.eval_prox)
                applies = np.repeat(np.array([p.imgIds[gr[1]] in chosen for gr in groups], dtype=bool), dt_sizes)
                t_app = torch.from_numpy(applies).to(device)
                prox = prox | ~t_app[t1]
            else:
                t_app = torch.ones(len(all_d), dtype=torch.bool, device=device)
            flat = torch.where(prox, flat, torch.full_like(flat, -1.0))      # a pair out of proximity can never be a match
            near = torch.zeros(len(all_d), dtype=torch.int32, device=device).index_add_(0, t1, prox.to(torch.int32)) > 0
            has_gt = torch.from_numpy(np.repeat(gt_sizes > 0, dt_sizes)).to(device)
            far = ~near & has_gt & t_app
        m = evaluate_groups(flat, dt_sizes, gt_sizes, torch.tensor([int(x[flag]) for x in all_g], dtype=torch.int32, device=device),
                            f32([x[rng_key] for x in all_g], (-1,)), f32([x[rng_key] for x in all_d], (-1,)), p.areaRng, p.iouThrs)
        if far is not None:
            m["dt_ignore"] = (m["dt_ignore"].bool() | far.view(1, 1, -1)).to(m["dt_ignore"].dtype)
        self._dev = {"groups": groups, "dt_sizes": dt_sizes, "gt_sizes": gt_sizes, "match": m, "device": device,
                     "scores": np.array([x["score"] for x in all_d], dtype=np.float64),
                     "dt_ids": np.array([x.get("id", 0) for x in all_d]), "gt_ids": np.array([x.get("id", 0) for x in all_g])}
        self._paramsEval = copy.deepcopy(self.params)
        self._evalImgs = None

    @property
    def evalImgs(self):
        """the reference's per-(category, range, image) list of dicts (:1541-1551), materialised on demand from the device
        results -- accumulate() does not need it"""
        if self._evalImgs is None and self._dev is not None:
            p, d = self._paramsEval, self._dev
            m = {k: v.cpu().numpy() for k, v in d["match"].items()}
            doff, goff = np.concatenate([[0], np.cumsum(d["dt_sizes"])]), np.concatenate([[0], np.cumsum(d["gt_sizes"])])
            where = {(ki, ii): n for n, (ki, ii, _, _) in enumerate(d["groups"])}
            out = []
            for ki, cat in enumerate(p.catIds):
                for ai, aRng in enumerate(p.areaRng):
                    for ii, img in enumerate(p.imgIds):
                        n = where.get((ki, ii))
                        if n is None:
                            out.append(None)
                            continue
                        ds, gs = slice(doff[n], doff[n + 1]), slice(goff[n], goff[n + 1])
                        order = m["gt_order"][ai, gs]
                        gids, dids = d["gt_ids"][gs], d["dt_ids"][ds]
                        dtm = m["dt_match"][ai][:, ds]
                        gtm = m["gt_match"][ai][:, gs][:, order]
                        out.append({"image_id": img, "category_id": cat, "aRng": aRng, "maxDet": p.maxDets[-1],
                                    "dtIds": list(dids), "gtIds": list(gids[order]),
                                    "dtMatches": np.where(dtm >= 0, gids[np.clip(dtm, 0, None)] if len(gids) else 0, 0).astype(np.float64),
                                    "gtMatches": np.where(gtm >= 0, dids[np.clip(gtm, 0, None)] if len(dids) else 0, 0).astype(np.float64),
                                    "dtScores": list(d["scores"][ds]), "gtIgnore": m["gt_ignore"][ai, gs][order].astype(np.float64),
                                    "dtIgnore": m["dt_ignore"][ai][:, ds].astype(bool)})
            self._evalImgs = out
        return self._evalImgs

    # ---- accumulate (:1172-1313) --------------------------------------------------------------------------------------
    def accumulate(self, p=None):
        assert self._dev is not None, "Please run evaluate() first"
        if p is None:
            p = self.params
        pe, d = self._paramsEval, self._dev
        if list(p.catIds) != list(pe.catIds) or list(map(tuple, p.areaRng)) != list(map(tuple, pe.areaRng)) or list(p.maxDets) != list(pe.maxDets) \
                or list(p.imgIds) != list(pe.imgIds):
            raise NotImplementedError("accumulate() with parameters other than evaluate()'s is not built on the device path")
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        dev = d["device"]
        groups, dt_sizes, gt_sizes = d["groups"], d["dt_sizes"], d["gt_sizes"]
        sumD = int(dt_sizes.sum())
        cat_of_group = np.array([g[0] for g in groups], dtype=np.int64)
        det_cat = np.repeat(cat_of_group, dt_sizes)
        det_rank = np.concatenate([np.arange(n) for n in dt_sizes]).astype(np.int32) if len(groups) else np.zeros(0, np.int32)
        # merge order: stable by -score inside a category, categories ascending; ties keep (image, in-image) order (:1250-1253)
        order = np.lexsort((np.arange(sumD), -d["scores"], det_cat)).astype(np.int32) if sumD else np.zeros(0, np.int32)
        cat_off = np.concatenate([[0], np.cumsum(np.bincount(det_cat, minlength=K))]).astype(np.int32)
        gt_ig = d["match"]["gt_ignore"].cpu().numpy()                                   # (A, sumG)
        gcat = np.repeat(cat_of_group, gt_sizes)
        npig = np.zeros((K, A), np.int32)
        for a in range(A):
            npig[:, a] = np.bincount(gcat[gt_ig[a] == 0], minlength=K)
        has_e = np.bincount(cat_of_group, minlength=K).astype(np.int32).clip(max=1)
        tod = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)               # noqa: E731
        prec = torch.full((T, R, K, A, M), -1.0, dtype=torch.float64, device=dev)
        rec = torch.full((T, K, A, M), -1.0, dtype=torch.float64, device=dev)
        scr = torch.full((T, R, K, A, M), -1.0, dtype=torch.float64, device=dev)
        L = iou3d._lib.get()
        _p = iou3d._lib.ptr
        t_order, t_off, t_rank, t_sc = tod(order), tod(cat_off), tod(det_rank), tod(d["scores"])
        t_npig, t_has, t_thr, t_md = tod(npig), tod(has_e), tod(np.asarray(p.recThrs, dtype=np.float64)), tod(np.asarray(p.maxDets, dtype=np.int32))
        dm, dg = d["match"]["dt_match"].contiguous(), d["match"]["dt_ignore"].contiguous()
        L.call("omni_eval_accumulate", _p(t_order), _p(t_off), _p(t_rank), _p(t_sc), _p(dm), _p(dg), _p(t_npig), _p(t_has), _p(t_thr),
               _p(t_md), K, A, M, T, R, sumD, _p(prec), _p(rec), _p(scr), iou3d._lib.stream_of(prec))
        self.eval = {"params": p, "counts": [T, R, K, A, M], "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"),
                     "precision": prec.cpu().numpy(), "recall": rec.cpu().numpy(), "scores": scr.cpu().numpy()}

    # ---- summarize (:1553-1704) ---------------------------------------------------------------------------------------
    def summarize(self):
        if not self.eval:
            raise Exception("Please run accumulate() first")
        p, ev, mode = self.params, self.eval, self.mode
        lines = []

        def one(ap=1, iouThr=None, areaRng="all", maxDets=100):
            fmt = (" {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}" if mode == "2D"
                   else " {:<18} {} @[ IoU={:<9} | depth={:>6s} | maxDets={:>3d} ] = {:0.3f}")
            iouStr = "{:0.2f}:{:0.2f}".format(p.iouThrs[0], p.iouThrs[-1]) if iouThr is None else "{:0.2f}".format(iouThr)
            aind = [i for i, a in enumerate(p.areaRngLbl) if a == areaRng]
            mind = [i for i, m in enumerate(p.maxDets) if m == maxDets]
            if ap == 1:
                s = ev["precision"]
                if iouThr is not None:
                    s = s[np.where(np.isclose(iouThr, p.iouThrs.astype(float)))[0]]
                s = s[:, :, :, aind, mind]
            else:
                s = ev["recall"]
                if iouThr is not None:
                    s = s[np.where(iouThr == p.iouThrs)[0]]
                s = s[:, :, aind, mind]
            mean_s = -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])
            lines.append("mode={} ".format(mode) + fmt.format("Average Precision" if ap == 1 else "Average Recall", "(AP)" if ap == 1 else "(AR)",
                                                             iouStr, areaRng, maxDets, mean_s))
            return mean_s

        thres = [0.5, 0.75, 0.95] if mode == "2D" else [0.15, 0.25, 0.50]
        L, md = p.areaRngLbl, p.maxDets
        stats = np.zeros((13,))
        stats[0] = one(1)
        for i in range(3):
            stats[1 + i] = one(1, iouThr=thres[i], maxDets=md[2])
        for i in range(3):
            stats[4 + i] = one(1, areaRng=L[1 + i], maxDets=md[2])
        for i in range(3):
            stats[7 + i] = one(0, maxDets=md[i])
        for i in range(3):
            stats[10 + i] = one(0, areaRng=L[1 + i], maxDets=md[2])
        self.stats = stats
        return "\n".join(lines)

    def __str__(self):
        self.summarize()


# =====================================================================================================================
# prediction plumbing around it (f-3): instances -> COCO-style records, the inference loop, the evaluator object
# =====================================================================================================================
def instances_to_coco_json(instances, img_id):
    """omni3d_evaluation.py:970-1013.  The reference converts each field with its own `.tolist()` after a per-field device
    copy and averages the corner depths one box at a time in numpy; here one host copy per field and a vectorised depth."""
    n = len(instances)
    if n == 0:
        return []
    cpu = instances.to("cpu") if instances.pred_boxes.tensor.is_cuda else instances
    xyxy = cpu.pred_boxes.tensor.numpy()
    boxes = np.concatenate([xyxy[:, :2], xyxy[:, 2:] - xyxy[:, :2]], axis=1).tolist()          # XYXY_ABS -> XYWH_ABS
    scores, classes = cpu.scores.tolist(), cpu.pred_classes.tolist()
    if cpu.has("pred_bbox3D"):
        b3 = cpu.pred_bbox3D.numpy()
        depth = b3[:, :, 2].mean(axis=1).tolist()
        bbox3D, center_cam, center_2D = b3.tolist(), cpu.pred_center_cam.tolist(), cpu.pred_center_2D.tolist()
        dimensions, pose = cpu.pred_dimensions.tolist(), cpu.pred_pose.tolist()
    else:
        bbox3D, center_cam, center_2D = np.ones([n, 8, 3]).tolist(), np.ones([n, 3]).tolist(), np.ones([n, 2]).tolist()
        dimensions, pose, depth = np.ones([n, 3]).tolist(), np.ones([n, 3, 3]).tolist(), [1.0] * n
    return [{"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k], "depth": depth[k], "bbox3D": bbox3D[k],
             "center_cam": center_cam[k], "center_2D": center_2D[k], "dimensions": dimensions[k], "pose": pose[k]} for k in range(n)]


def inference_on_dataset(model, data_loader):
    """omni3d_evaluation.py:522-640 without the timing / logging: model in eval mode over the loader -> list of
    {'image_id', 'K', 'width', 'height', 'instances': [COCO-style records]}"""
    was_training = model.training
    model.eval()
    out = []
    with torch.no_grad():
        for inputs in data_loader:
            outputs = model(inputs)
            for inp, o in zip(inputs, outputs):
                out.append({"image_id": inp["image_id"], "K": inp["K"], "width": inp["width"], "height": inp["height"],
                            "instances": instances_to_coco_json(o["instances"], inp["image_id"])})
    model.train(was_training)
    return out


_METRICS = {"2D": ["AP", "AP50", "AP75", "AP95", "APs", "APm", "APl"], "3D": ["AP", "AP15", "AP25", "AP50", "APn", "APm", "APf"]}


def _derive_results(ev, mode, class_names):
    """omni3d_evaluation.py:765-846: the seven headline numbers (x100, NaN when undefined) + per-category AP ('AP-<name>': mean
    of the precision table over thresholds and recall points at area 'all', maxDets 100)"""
    res = {name: float(ev.stats[i] * 100 if ev.stats[i] >= 0 else "nan") for i, name in enumerate(_METRICS[mode])}
    if class_names is None or len(class_names) <= 1:
        return res
    prec = ev.eval["precision"]
    assert len(class_names) == prec.shape[2], (len(class_names), prec.shape)
    for k, name in enumerate(class_names):
        vals = prec[:, :, k, 0, -1]
        vals = vals[vals > -1]
        res["AP-" + name] = float(np.mean(vals) * 100) if vals.size else float("nan")
    return res


class Omni3DEvaluator:
    """Per-dataset evaluator (reference :643-935, a detectron2 COCOEvaluator subclass).

    Reference form: `Omni3DEvaluator(dataset_name, output_dir=..., filter_settings=..., only_2d=..., eval_prox=..., distributed=...)`
    -- the ground truth is the dataset's registered annotation file read through `Omni3D([json_file], filter_settings)`;
    predictions are per-image dicts {'image_id', 'K', 'width', 'height', 'instances': [records with CONTIGUOUS category ids]}.
    `evaluate()` maps the categories back to dataset ids, keeps the dataset's own categories, writes
    `omni_instances_results.json`, runs Omni3Deval in 2D (and 3D) and returns {'bbox_2D': {...}, 'bbox_3D': {...},
    'log_str_2D', 'log_str_3D', 'bbox_*_merge'} ('*_merge' = what the helper needs to score the union of several datasets; the
    reference caches per-image match tables under '*_evals_per_cat_area' for the same purpose).

    Short form (in-memory ground truth): `Omni3DEvaluator(gt_annotations, img_ids, cat_ids, only_2d)` with
    `process(inputs, outputs)` taking model outputs -> {'bbox': {'AP2D', 'AP3D', 'omni_eval_*'}}."""

    def __init__(self, dataset_name, tasks=None, distributed=True, output_dir=None, *, max_dets_per_image=None, use_fast_impl=False,
                 eval_prox=False, only_2d=False, filter_settings=None, img_ids=None, cat_ids=None):
        self._only_2d, self._eval_prox, self._output_dir, self._distributed = only_2d, eval_prox, output_dir, distributed
        if not isinstance(dataset_name, str):                   # short form: (gt_annotations, img_ids, cat_ids, only_2d)
            self._gt, self._omni_api = dataset_name, None
            self._img_ids = tasks if tasks is not None else img_ids
            self._cat_ids = cat_ids if isinstance(distributed, bool) else distributed
            if output_dir is not None and not isinstance(output_dir, str):
                self._only_2d, self._output_dir = bool(output_dir), None
            self.reset()
            return
        from ...d2.data import MetadataCatalog
        from ..data.datasets import Omni3D
        self._filter_settings = filter_settings if filter_settings is not None else {}
        self._metadata = MetadataCatalog.get(dataset_name)
        self._omni_api = Omni3D([self._metadata.json_file], self._filter_settings)
        self._do_evaluation = "annotations" in self._omni_api.dataset
        self.reset()

    def reset(self):
        self._predictions = []
        self._results = {}

    def process(self, inputs, outputs):
        for inp, o in zip(inputs, outputs):
            recs = o["instances"] if isinstance(o["instances"], list) else instances_to_coco_json(o["instances"], inp["image_id"])
            if self._omni_api is None:
                self._predictions.extend(recs)
            else:
                pred = {"image_id": inp["image_id"], "K": inp["K"], "width": inp["width"], "height": inp["height"], "instances": recs}
                if "p2" in inp:
                    pred["p2"] = inp["p2"]
                self._predictions.append(pred)

    def _evaluate_short(self):
        res = {}
        for mode in (["2D"] if self._only_2d else ["2D", "3D"]):
            ev = Omni3Deval(AnnotationIndex(copy.deepcopy(self._gt), self._img_ids, self._cat_ids),
                            AnnotationIndex(copy.deepcopy(self._predictions), self._img_ids, self._cat_ids), mode=mode)
            ev.evaluate()
            ev.accumulate()
            ev.summarize()
            res["AP" + mode] = float(ev.stats[0] * 100)
            res["omni_eval_" + mode] = ev
        return {"bbox": res}

    def evaluate(self, img_ids=None):
        if self._omni_api is None:
            return self._evaluate_short()
        from ...d2.data import MetadataCatalog
        predictions = self._predictions
        if self._distributed:
            from ...d2 import comm
            comm.synchronize()
            gathered = comm.gather(predictions, dst=0)
            predictions = [x for part in gathered for x in part]
            if not comm.is_main_process():
                return {}
        self._results = {}
        if len(predictions) == 0:
            logging.getLogger(__name__).warning("[Omni3DEvaluator] Did not receive valid predictions.")
            return {}
        if self._output_dir:
            os.makedirs(self._output_dir, exist_ok=True)
            torch.save(predictions, os.path.join(self._output_dir, "instances_predictions.pth"))
        model_meta = MetadataCatalog.get("omni3d_model")
        model_classes = model_meta.thing_classes
        # the split's own tables are filled in when its dataset dicts are loaded (load_omni3d_json); without a loader run the
        # model's id table and the categories of the annotation file stand in (they are what the loader would have stored)
        id_map = self._metadata.get("thing_dataset_id_to_contiguous_id") or model_meta.thing_dataset_id_to_contiguous_id
        split_classes = self._metadata.get("thing_classes") or [c["name"] for c in sorted(self._omni_api.dataset["categories"], key=lambda c: c["id"])]
        to_dataset_id = {v: k for k, v in id_map.items()}
        num_classes = len(to_dataset_id)
        kept = []
        for rec in (r for pred in predictions for r in pred["instances"]):
            c = rec["category_id"]
            assert c < num_classes, f"A prediction has class={c}, but the model only has {num_classes} classes"
            if model_classes[c] in split_classes:                         # categories this dataset is annotated for
                r = dict(rec)
                r["category_id"] = to_dataset_id[c]
                kept.append(r)
        if self._output_dir:
            with open(os.path.join(self._output_dir, "omni_instances_results.json"), "w") as f:
                json.dump(kept, f)
        if not self._do_evaluation or len(kept) == 0:
            return copy.deepcopy(self._results)
        omni_dt = self._omni_api.loadRes(kept)
        for mode in (["2D"] if self._only_2d else ["2D", "3D"]):
            ev = Omni3Deval(self._omni_api, omni_dt, mode=mode, eval_prox=self._eval_prox)
            if img_ids is not None:
                ev.params.imgIds = img_ids
            ev.evaluate()
            ev.accumulate()
            self._results["log_str_" + mode] = ev.summarize()
            self._results["bbox_" + mode] = _derive_results(ev, mode, split_classes)
            self._results["bbox_" + mode + "_merge"] = {"gt": self._omni_api, "dt": kept, "eval_prox": self._eval_prox,
                                                        "img_ids": list(ev.params.imgIds), "cat_ids": list(ev.params.catIds)}
        return copy.deepcopy({k: v for k, v in self._results.items() if not k.endswith("_merge")}) | \
            {k: v for k, v in self._results.items() if k.endswith("_merge")}


class Omni3DEvaluationHelper:
    """omni3d_evaluation.py:168-520: one `Omni3DEvaluator` per test split, their printed tables, and `summarize_all()` = the
    metrics of the union of all splits (<Concat>) plus the Omni3D / Omni3D_In / Omni3D_Out aggregates.  Needs
    `MetadataCatalog.get('omni3d_model').{thing_classes, thing_dataset_id_to_contiguous_id}`.  The union is scored by one
    evaluation over the concatenated ground truth and detections (images of different splits are disjoint, so this equals the
    reference's concatenation of cached per-image match tables), with proximity evaluation applied to the images of the splits
    that use it."""

    def __init__(self, dataset_names, filter_settings, output_folder, iter_label="-", only_2d=False):
        from collections import OrderedDict
        from ...d2.data import MetadataCatalog
        from ..data.datasets import simple_register
        self.dataset_names, self.filter_settings, self.output_folder = list(dataset_names), filter_settings, output_folder
        self.iter_label, self.only_2d = iter_label, only_2d
        self.evaluators, self.results = OrderedDict(), OrderedDict()
        self.results_analysis, self.results_omni3d = OrderedDict(), OrderedDict()
        self.overall_imgIds, self.overall_catIds = set(), set()
        self.output_folders = {n: os.path.join(output_folder, n) for n in self.dataset_names}
        for name in self.dataset_names:
            if MetadataCatalog.get(name).get("json_file") is None:
                simple_register(name, filter_settings, filter_empty=False)
            ev = Omni3DEvaluator(name, output_dir=self.output_folders[name], filter_settings=filter_settings, only_2d=only_2d,
                                 eval_prox=("Objectron" in name or "SUNRGBD" in name), distributed=False)
            ev.reset()
            self.evaluators[name] = ev
            self.overall_imgIds.update(ev._omni_api.getImgIds())
            self.overall_catIds.update(ev._omni_api.getCatIds())

    def add_predictions(self, dataset_name, predictions):
        self.evaluators[dataset_name]._predictions += predictions

    def save_predictions(self, dataset_name):
        os.makedirs(self.output_folders[dataset_name], exist_ok=True)
        torch.save(self.evaluators[dataset_name]._predictions, os.path.join(self.output_folders[dataset_name], "instances_predictions.pth"))

    @staticmethod
    def _mean(values):
        values = list(values)
        return float(np.mean(values)) if values else float("nan")

    def _aggregates(self, res2d, res3d, categories):
        nan = float("nan")
        out = {"AP2D": self._mean(res2d["AP-" + c] for c in categories), "AP3D": nan}
        if not self.only_2d:
            out["AP3D"] = self._mean(res3d["AP-" + c] for c in categories)
        return out

    def evaluate(self, dataset_name):
        from ..data.datasets import get_omni3d_categories
        from ..vis import logperf
        log = logging.getLogger(__name__)
        if dataset_name not in self.results:
            self.results[dataset_name] = self.evaluators[dataset_name].evaluate()
        res = self.results[dataset_name]
        tag = "{} iter={} mode=".format(dataset_name, self.iter_label)
        log.info("\n" + res["log_str_2D"].replace("mode=2D", tag + "2D"))
        if not self.only_2d:
            log.info("\n" + res["log_str_3D"].replace("mode=3D", tag + "3D"))
        names = self.filter_settings["category_names"]
        r2, r3 = res["bbox_2D"], res.get("bbox_3D", {})
        present = {c for c in names if "AP-" + c in r2}
        general = self._aggregates(r2, r3, present)
        omni = {"AP2D": float("nan"), "AP3D": float("nan")}
        split_cats = get_omni3d_categories(dataset_name)
        if len(split_cats - present) == 0:
            omni = self._aggregates(r2, r3, split_cats)
        self.results_omni3d[dataset_name] = {"iters": self.iter_label, **omni}
        extras = {k: (r3[k] if not self.only_2d else float("nan")) for k in ("AP15", "AP25", "AP50", "APn", "APm", "APf")}
        self.results_analysis[dataset_name] = {"iters": self.iter_label, "AP2D": general["AP2D"], "AP3D": general["AP3D"],
                                               "AP3D@15": extras["AP15"], "AP3D@25": extras["AP25"], "AP3D@50": extras["AP50"],
                                               "AP3D-N": extras["APn"], "AP3D-M": extras["APm"], "AP3D-F": extras["APf"]}
        logperf.print_ap_category_histogram(dataset_name, self._per_category(r2, r3))

    def _per_category(self, r2, r3):
        from collections import OrderedDict
        out = OrderedDict()
        for c in self.filter_settings["category_names"]:
            a2 = r2.get("AP-" + c, float("nan"))
            a3 = r3.get("AP-" + c, float("nan")) if not self.only_2d else float("nan")
            if not np.isnan(a2) or not np.isnan(a3):
                out[c] = {"AP2D": a2, "AP3D": a3}
        return out

    def summarize_all(self):
        from ...d2.data import MetadataCatalog
        from ..data.datasets import get_omni3d_categories
        from ..vis import logperf
        for name in self.dataset_names:
            if name not in self.results:
                self.evaluate(name)
        meta = MetadataCatalog.get("omni3d_model")
        cat_ids = sorted(self.overall_catIds)
        ordered = [meta.thing_classes[meta.thing_dataset_id_to_contiguous_id[c]] for c in cat_ids]
        categories = set(ordered)
        merged = {}
        for mode in (["2D"] if self.only_2d else ["2D", "3D"]):
            gts, dts, prox_imgs = [], [], set()
            for name in self.dataset_names:
                rec = self.results[name].get("bbox_" + mode + "_merge")
                if rec is None:
                    continue
                gts += rec["gt"].loadAnns(rec["gt"].getAnnIds(imgIds=rec["img_ids"], catIds=rec["cat_ids"]))
                dts += rec["dt"]
                if rec["eval_prox"]:
                    prox_imgs.update(rec["img_ids"])
            ev = Omni3Deval(AnnotationIndex(copy.deepcopy(gts), self.overall_imgIds, cat_ids),
                            AnnotationIndex(copy.deepcopy(dts), self.overall_imgIds, cat_ids), mode=mode,
                            eval_prox=(prox_imgs if prox_imgs else False))
            ev.evaluate()
            ev.accumulate()
            ev.summarize()
            merged[mode] = _derive_results(ev, mode, ordered if len(ordered) > 1 else None)
            if len(ordered) == 1:       # _derive_results skips the per-category part for a single class
                merged[mode]["AP-" + ordered[0]] = merged[mode]["AP"]
        r2, r3 = merged["2D"], merged.get("3D", {})
        general = self._aggregates(r2, r3, categories)
        extras = {k: (r3[k] if not self.only_2d else float("nan")) for k in ("AP15", "AP25", "AP50", "APn", "APm", "APf")}
        self.results_analysis["<Concat>"] = {"iters": self.iter_label, "AP2D": general["AP2D"], "AP3D": general["AP3D"],
                                             "AP3D@15": extras["AP15"], "AP3D@25": extras["AP25"], "AP3D@50": extras["AP50"],
                                             "AP3D-N": extras["APn"], "AP3D-M": extras["APm"], "AP3D-F": extras["APf"]}
        for label, key in (("Omni3D_Out", "omni3d_out"), ("Omni3D_In", "omni3d_in"), ("Omni3D", "omni3d")):
            want = get_omni3d_categories(key)
            agg = self._aggregates(r2, r3, want) if len(want - categories) == 0 else {"AP2D": float("nan"), "AP3D": float("nan")}
            self.results_omni3d[label] = {"iters": self.iter_label, **agg}
        logperf.print_ap_category_histogram("<Concat>", self._per_category(r2, r3))
        logperf.print_ap_analysis_histogram(self.results_analysis)
        logperf.print_ap_omni_histogram(self.results_omni3d)
        return self.results_analysis, self.results_omni3d
