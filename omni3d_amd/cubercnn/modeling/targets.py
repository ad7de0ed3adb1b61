"""Packing of per-image ground truth into flat device arrays (one H2D copy per step).

The reference moves every `Instances` field to the device separately and then splits valid /
ignore GT with boolean masks on the GPU (rpn.py:48-49, roi_heads.py:866-867), each a host sync.
Here the split is done on the host copy that the data loader produced anyway."""
import numpy as np
import torch


MAX_GT_PER_IMAGE = 1024     # csrc/rpn_roi.hip MAXG (with MODEL.RPN.POST_NMS_TOPK_TRAIN 1000 the sampler's 2048 candidate slots still hold proposals + appended GT)


class PackedTargets:
    """gt (G,4), gt_cls (G) int32, gt_off (B+1) int32 -- valid GT; ign (Gi,4), ign_off (B+1);
    gt3d (G,9) gt_boxes3D rows, gtpose (G,9); Ks (B,4) = [fx,fy,cx,cy]/ratio; v2r (B); ratio (B); image_hw (B,2)."""

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device, non_blocking=True))
        return self


def pack_targets(batched_inputs, image_sizes, virtual_focal=512.0, with_gt=True):
    """from the model's input dicts (rcnn3d.py:41-56): instances, K and the original height of every image"""
    insts = [info.get("instances") if with_gt else None for info in batched_inputs]
    Ks = [info["K"] for info in batched_inputs]
    ratios = [info["height"] / h_net for info, (h_net, _) in zip(batched_inputs, image_sizes)]      # rcnn3d.py:50 im_scales_ratio
    return pack_instances(insts, image_sizes, Ks, ratios, virtual_focal)


_cache = {"insts": None, "elems": [], "sizes": None, "Ks": None, "ratios": None, "val": None}


def _same_scalars(a, b):
    """Ks / ratios of two calls: equal values (they are a handful of floats per image; never trust identity here)"""
    if a is None or b is None:
        return a is None and b is None
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)
        y = y.detach().cpu().numpy() if isinstance(y, torch.Tensor) else np.asarray(y, dtype=np.float64)
        if x.shape != y.shape or not np.array_equal(x, y):
            return False
    return True


def pack_instances_cached(insts, image_sizes, Ks=None, ratios=None, virtual_focal=512.0, device=None):
    """`pack_instances` memoised on the gt list OBJECT: the reference's RCNN3D.forward hands the same `gt_instances` list to the
    proposal generator (rcnn3d.py:65) and to the ROI heads (:70), so when this package's modules are driven through the
    reference's contracts (list[Instances]) the ground truth is still packed / copied to the device once per step.  The cache
    keeps a strong reference to that list and compares with `is` (CPython recycles the address of a dead list at once, so an
    id() key would hand the previous step's targets to a new list), every element must be the same object too (a caller may
    refill a list in place), and an entry made with intrinsics is only reused for EQUAL intrinsics / ratios.  The entry without
    intrinsics (RPN) is upgraded when the ROI heads supply Ks / ratios."""
    sizes = tuple(map(tuple, image_sizes))
    hit = None
    if (insts is not None and _cache["insts"] is insts and _cache["sizes"] == sizes and len(insts) == len(_cache["elems"])
            and all(a is b for a, b in zip(insts, _cache["elems"])) and any(a is not None for a in insts)):
        hit = _cache["val"]
    if hit is not None:
        if Ks is None:
            return hit
        if hit.has_intrinsics and _same_scalars(Ks, _cache["Ks"]) and _same_scalars(ratios, _cache["ratios"]):
            return hit
    t = pack_instances(insts, image_sizes, Ks, ratios, virtual_focal)
    if device is not None:
        t = t.to(device)
    _cache.update(insts=insts, elems=list(insts) if insts is not None else [], sizes=sizes, Ks=Ks, ratios=ratios, val=t)
    return t


def pack_instances(insts, image_sizes, Ks=None, ratios=None, virtual_focal=512.0):
    """insts: list of Instances (gt_classes, gt_boxes[, gt_boxes3D, gt_poses]) or None per image; Ks: per-image 3x3
    intrinsics of the ORIGINAL image (list / tensor) or None; ratios: original height / network height per image."""
    B = len(image_sizes)
    t = PackedTargets()
    gt, cls, g3, gp, ign = [], [], [], [], []
    goff, ioff = [0], [0]
    Kp, v2r, ratio = [], [], []
    t.has_intrinsics = Ks is not None
    for n, (h_net, w_net) in enumerate(image_sizes):
        r = float(ratios[n]) if ratios is not None else 1.0
        if Ks is not None:
            K = np.asarray(Ks[n].cpu() if isinstance(Ks[n], torch.Tensor) else Ks[n], dtype=np.float64)
            Kp.append([K[0, 0] / r, K[1, 1] / r, K[0, 2] / r, K[1, 2] / r])      # roi_heads.py:374-378
            v2r.append((h_net * K[1, 1]) / (virtual_focal * (h_net * r)))        # roi_heads.py:396-403, math_util.py:581-592
        else:
            Kp.append([1.0, 1.0, 0.0, 0.0])
            v2r.append(1.0)
        ratio.append(r)
        n_valid = n_ign = 0
        inst = insts[n] if insts is not None else None
        if inst is not None and inst.has("gt_classes"):
            c = inst.gt_classes.cpu().numpy()
            b = inst.gt_boxes.tensor.cpu().numpy().reshape(-1, 4)
            ok = c >= 0
            n_valid, n_ign = int(ok.sum()), int((~ok).sum())
            gt.append(b[ok]); cls.append(c[ok]); ign.append(b[~ok])
            g3.append(inst.gt_boxes3D.cpu().numpy().reshape(-1, 9)[ok] if inst.has("gt_boxes3D") else np.zeros((n_valid, 9)))
            gp.append(inst.gt_poses.cpu().numpy().reshape(-1, 9)[ok] if inst.has("gt_poses") else np.zeros((n_valid, 9)))
        if n_valid > MAX_GT_PER_IMAGE or n_ign > MAX_GT_PER_IMAGE:
            # the matching / sampling kernels keep an image's ground truth in LDS (csrc/rpn_roi.hip MAXG); the reference has
            # no such cap, so fail loudly instead of training on truncated targets
            raise ValueError(f"image {n}: {n_valid} valid / {n_ign} ignore boxes exceed the kernels' capacity of {MAX_GT_PER_IMAGE} per image")
        goff.append(goff[-1] + n_valid)
        ioff.append(ioff[-1] + n_ign)
    Ks = Kp

    def cat(lst, w, dt):
        a = np.concatenate(lst, 0) if lst else np.zeros((0, w))
        a = a.reshape(-1, w) if w > 1 else a.reshape(-1)
        if a.shape[0] == 0:     # kernels are handed a valid pointer even when a list is empty
            a = np.zeros((1, w) if w > 1 else (1,))
        return torch.from_numpy(np.ascontiguousarray(a.astype(dt)))

    t.num_gt, t.num_ign = goff[-1], ioff[-1]
    t.gt, t.gt_cls = cat(gt, 4, np.float32), cat(cls, 1, np.int32)
    t.gt3d, t.gtpose, t.ign = cat(g3, 9, np.float32), cat(gp, 9, np.float32), cat(ign, 4, np.float32)
    t.gt_off = torch.tensor(goff, dtype=torch.int32)
    t.ign_off = torch.tensor(ioff, dtype=torch.int32)
    t.Ks = torch.tensor(Ks, dtype=torch.float32)
    t.v2r = torch.tensor(v2r, dtype=torch.float32)
    t.ratio = torch.tensor(ratio, dtype=torch.float32)
    t.image_hw = torch.tensor([[h, w] for h, w in image_sizes], dtype=torch.int32)
    t.B = B
    return t
