"""Packing of per-image ground truth into flat device arrays (one H2D copy per step).

The reference moves every `Instances` field to the device separately and then splits valid /
ignore GT with boolean masks on the GPU (rpn.py:48-49, roi_heads.py:866-867), each a host sync.
Here the split is done on the host copy that the data loader produced anyway."""
import numpy as np
import torch


class PackedTargets:
    """gt (G,4), gt_cls (G) int32, gt_off (B+1) int32 -- valid GT; ign (Gi,4), ign_off (B+1);
    gt3d (G,9) gt_boxes3D rows, gtpose (G,9); Ks (B,4) = [fx,fy,cx,cy]/ratio; v2r (B); ratio (B); image_hw (B,2)."""

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device, non_blocking=True))
        return self


def pack_targets(batched_inputs, image_sizes, virtual_focal=512.0, with_gt=True):
    B = len(batched_inputs)
    t = PackedTargets()
    gt, cls, g3, gp, ign = [], [], [], [], []
    goff, ioff = [0], [0]
    Ks, v2r, ratio = [], [], []
    for info, (h_net, w_net) in zip(batched_inputs, image_sizes):
        r = info["height"] / h_net                                  # rcnn3d.py:50 im_scales_ratio
        K = np.asarray(info["K"], dtype=np.float64)
        Ks.append([K[0, 0] / r, K[1, 1] / r, K[0, 2] / r, K[1, 2] / r])      # roi_heads.py:374-378
        v2r.append((h_net * K[1, 1]) / (virtual_focal * (h_net * r)))        # roi_heads.py:396-403, math_util.py:581-592
        ratio.append(r)
        n_valid = n_ign = 0
        if with_gt and "instances" in info:
            inst = info["instances"]
            c = inst.gt_classes.cpu().numpy()
            b = inst.gt_boxes.tensor.cpu().numpy().reshape(-1, 4)
            ok = c >= 0
            n_valid, n_ign = int(ok.sum()), int((~ok).sum())
            gt.append(b[ok]); cls.append(c[ok]); ign.append(b[~ok])
            g3.append(inst.gt_boxes3D.cpu().numpy().reshape(-1, 9)[ok] if inst.has("gt_boxes3D") else np.zeros((n_valid, 9)))
            gp.append(inst.gt_poses.cpu().numpy().reshape(-1, 9)[ok] if inst.has("gt_poses") else np.zeros((n_valid, 9)))
        goff.append(goff[-1] + n_valid)
        ioff.append(ioff[-1] + n_ign)

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
