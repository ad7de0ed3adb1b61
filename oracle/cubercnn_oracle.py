"""oracle/cubercnn_oracle.py -- TEST INFRASTRUCTURE (CPU oracle).  Never imported by the product.

Plain-PyTorch (CPU fp32, autograd) restatement of the reference's OWN hot-path arithmetic
(/root/reference/cubercnn/modeling/...), written functionally so that every random choice is an
explicit input: `torch.multinomial(w, n)` (rpn.py:318,322) is `topk(w / E, n)` with E ~ Exp(1) --
exactly torch's algorithm for sampling without replacement -- and E is injected by the caller, so
the HIP kernels and this oracle can be driven by the same variates.

It travels to the GPU box (the reference checkout does not).  In the build container
tests/test_reference_pin.py checks these functions against the reference files themselves run
under oracle/ref_harness.py; tests/golden/ holds the resulting fixtures.
"""
import math

import torch
import torch.nn.functional as F

from oracle import upstream as U
from omni3d_amd.d2.structures import Boxes

SQRT_2 = 1.41421356


# ---------------------------------------------------------------------------------------------------
# proposal_generator/rpn.py
# ---------------------------------------------------------------------------------------------------

def subsample_labels_E(labels, num_samples, positive_fraction, bg_label, matched_ious, E, eps=1e-4):
    """rpn.py:275-328 with the multinomial's exponential variates E (same length as labels) injected."""
    positive = U.nonzero_tuple((labels != -1) & (labels != bg_label))[0]
    negative = U.nonzero_tuple(labels == bg_label)[0]
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    perm1 = ((matched_ious[positive] + eps) / E[positive]).topk(num_pos)[1] if num_pos > 0 else positive.new_zeros(0)
    perm2 = ((matched_ious[negative] + eps) / E[negative]).topk(num_neg)[1] if num_neg > 0 else negative.new_zeros(0)
    return positive[perm1], negative[perm2]


def rpn_label_and_sample(anchors, gt_boxes, ign_boxes, E, thresholds=(0.05, 0.05), match_labels=(0, -1, 1),
                         batch_size_per_image=256, positive_fraction=1.0, ignore_thresh=0.5):
    """RPNWithIgnore.label_and_sample_anchors for ONE image (rpn.py:56-108).
    anchors (A,4), gt_boxes (M,4) valid GT, ign_boxes (Mi,4) -> labels (A,) int8 in {-1,0,1},
    matched_idxs (A,), matched_ious (A,)."""
    matcher = U.Matcher(list(thresholds), list(match_labels), allow_low_quality_matches=True)
    mqm = U.pairwise_iou(Boxes(gt_boxes), Boxes(anchors))
    matched_idxs, gt_labels_i = matcher(mqm)
    matched_ious = mqm[matched_idxs, torch.arange(mqm.shape[1])]
    _, best_ious_gt_ind = mqm.max(dim=1)
    best_inds = torch.tensor(sorted(set(best_ious_gt_ind.tolist()) & set((gt_labels_i == 1).nonzero().squeeze(1).tolist())),
                             dtype=torch.int64)
    pos_idx, neg_idx = subsample_labels_E(gt_labels_i, batch_size_per_image, positive_fraction, 0, matched_ious, E)
    labels = torch.full_like(gt_labels_i, -1)
    labels[pos_idx] = 1
    labels[neg_idx] = 0
    if best_inds.numel() > 0:
        labels[best_inds] = 1
    if len(ign_boxes) > 0:
        background_inds = (labels == 0).nonzero().squeeze()
        if background_inds.numel() > 1:
            ioa = U.pairwise_ioa(Boxes(ign_boxes), Boxes(anchors[background_inds]))
            labels[background_inds[ioa.max(0)[0] >= ignore_thresh]] = -1
    return labels, matched_idxs, matched_ious, (pos_idx, neg_idx)


def matched_pairwise_iou(b1, b2):
    """rpn.py:330-353"""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, :2], b2[:, :2])
    rb = torch.min(b1[:, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    return inter / (a1 + a2 - inter)


def rpn_losses_iouness(anchors, logits, deltas, labels, matched_gt_boxes, batch_size_per_image=256):
    """RPNWithIgnore.losses + _dense_box_regression_loss_with_uncertainty (rpn.py:129-273), 'IoUness'.
    anchors (A,4); logits (N,A); deltas (N,A,4); labels (N,A); matched_gt_boxes (N,A,4).
    -> {'rpn/cls','rpn/loc'}, stats dict."""
    N = logits.shape[0]
    fg = labels == 1
    b2b = U.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0))
    boxes_fg = anchors.unsqueeze(0).repeat(N, 1, 1)[fg]
    gt_fg = matched_gt_boxes[fg].detach()
    t = matched_pairwise_iou(boxes_fg, gt_fg).detach()
    loss_conf = (F.binary_cross_entropy_with_logits(logits[fg], t, reduction="none") * t).sum()
    gt_deltas = torch.stack([b2b.get_deltas(anchors, k) for k in matched_gt_boxes])
    loss_reg = (U.smooth_l1_loss(deltas[fg], gt_deltas[fg], beta=0.0, reduction="none").sum(dim=1) * t).sum()
    normalizer = batch_size_per_image * N
    stats = {"rpn/num_pos_anchors": fg.sum().item() / N, "rpn/num_neg_anchors": (labels == 0).sum().item() / N,
             "rpn/conf_pos_anchors": torch.sigmoid(logits[fg]).mean().item(),
             "rpn/conf_neg_anchors": torch.sigmoid(logits[~fg]).mean().item()}
    return {"rpn/cls": loss_conf / normalizer, "rpn/loc": loss_reg / normalizer}, stats


def rpn_losses_plain(anchors, logits, deltas, labels, matched_gt_boxes, batch_size_per_image=256):
    """RPNWithIgnore.losses with OBJECTNESS_UNCERTAINTY 'none' (rpn.py:181-203): detectron2's `_dense_box_regression_loss`
    (smooth_l1, beta 0, summed over the foreground) and BCE-with-logits against the 0 / 1 labels over the valid anchors."""
    N = logits.shape[0]
    fg = labels == 1
    b2b = U.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0))
    loss_reg = U._dense_box_regression_loss([anchors], b2b, [deltas], [k for k in matched_gt_boxes], fg, box_reg_loss_type="smooth_l1",
                                            smooth_l1_beta=0.0)
    valid = labels >= 0
    loss_cls = F.binary_cross_entropy_with_logits(logits[valid], labels[valid].to(torch.float32), reduction="sum")
    normalizer = batch_size_per_image * N
    return {"rpn/cls": loss_cls / normalizer, "rpn/loc": loss_reg / normalizer}


# ---------------------------------------------------------------------------------------------------
# roi_heads/roi_heads.py: label_and_sample_proposals
# ---------------------------------------------------------------------------------------------------

def roi_label_and_sample(prop_boxes, gt_boxes, gt_classes, ign_boxes, E, num_classes=50, iou_thr=0.5,
                         batch_size_per_image=512, positive_fraction=0.25, ignore_thresh=0.5, append_gt=True):
    """ROIHeads3D.label_and_sample_proposals for ONE image (roi_heads.py:862-929).
    -> (boxes of sampled, classes, matched gt idx, sampled candidate indices)."""
    cand = torch.cat([prop_boxes, gt_boxes], dim=0) if append_gt else prop_boxes
    has_gt = len(gt_boxes) > 0
    mqm = U.pairwise_iou(Boxes(gt_boxes), Boxes(cand))
    matcher = U.Matcher([iou_thr], [0, 1], allow_low_quality_matches=False)
    matched_idxs, matched_labels = matcher(mqm)
    if len(ign_boxes) > 0:
        background_inds = (matched_labels == 0).nonzero().squeeze()
        if background_inds.numel() > 1:
            ioa = U.pairwise_ioa(Boxes(ign_boxes), Boxes(cand[background_inds]))
            matched_labels[background_inds[ioa.max(0)[0] >= ignore_thresh]] = -1
    if has_gt:
        matched_ious = mqm[matched_idxs, torch.arange(mqm.shape[1])]
        cls = gt_classes[matched_idxs].clone()
        cls[matched_labels == 0] = num_classes
        cls[matched_labels == -1] = -1
    else:
        matched_ious = torch.zeros(len(cand))
        cls = torch.zeros_like(matched_idxs) + num_classes
    fg_idx, bg_idx = subsample_labels_E(cls, batch_size_per_image, positive_fraction, num_classes, matched_ious, E[: len(cand)])
    sampled = torch.cat([fg_idx, bg_idx], dim=0)
    return cand[sampled], cls[sampled], matched_idxs[sampled], sampled


# ---------------------------------------------------------------------------------------------------
# roi_heads/fast_rcnn.py: FastRCNNOutputs.losses
# ---------------------------------------------------------------------------------------------------

def fast_rcnn_losses(scores, deltas, gt_classes, proposal_boxes, gt_boxes, num_classes, weights=(10.0, 10.0, 5.0, 5.0)):
    """fast_rcnn.py:145-194: CE mean + L1 on GT-class deltas of the foreground / #ROIs."""
    b2b = U.Box2BoxTransform(weights=weights)
    n = max(gt_classes.numel(), 1.0)
    loss_cls = U.cross_entropy(scores, gt_classes, reduction="mean")
    fg = U.nonzero_tuple((gt_classes >= 0) & (gt_classes < num_classes))[0]
    fg_pred = deltas.view(-1, num_classes, 4)[fg, gt_classes[fg]]
    gt_d = b2b.get_deltas(proposal_boxes[fg], gt_boxes[fg])
    loss_reg = U.smooth_l1_loss(fg_pred, gt_d, 0.0, reduction="none").sum() / n
    return {"BoxHead/loss_cls": loss_cls, "BoxHead/loss_box_reg": loss_reg}


# ---------------------------------------------------------------------------------------------------
# util/math_util.py pieces + roi_heads/cube_head.py + roi_heads.py:_forward_cube
# ---------------------------------------------------------------------------------------------------

def get_cuboid_verts(box3d, R):
    """math_util.py:116-219 (vertices only): box3d (n,6)=[X,Y,Z,W,H,L], R (n,3,3) -> (n,8,3)."""
    n = len(box3d)
    x3d, y3d, z3d, w3d, h3d, l3d = [box3d[:, i].unsqueeze(1) for i in range(6)]
    sx = torch.tensor([-1, 1, 1, -1, -1, 1, 1, -1.0])
    sy = torch.tensor([-1, -1, 1, 1, -1, -1, 1, 1.0])
    sz = torch.tensor([-1, -1, -1, -1, 1, 1, 1, 1.0])
    verts = torch.stack([l3d / 2 * sx, h3d / 2 * sy, w3d / 2 * sz], dim=1)      # (n,3,8)
    verts = R @ verts
    verts = verts + torch.stack([x3d, y3d, z3d], dim=1)
    return verts.transpose(1, 2)


def R_from_allocentric(K, R_view, u, v):
    """math_util.py:651-679 (tensor branch)."""
    fx, fy, sx, sy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    oray = torch.stack(((u - sx) / fx, (v - sy) / fy, torch.ones_like(u))).T
    oray = oray / torch.linalg.norm(oray, dim=1).unsqueeze(1)
    angle = torch.acos(oray[:, -1])
    axis = torch.zeros_like(oray)
    axis[:, 0] = axis[:, 0] - oray[:, 1]
    axis[:, 1] = axis[:, 1] + oray[:, 0]
    norms = torch.linalg.norm(axis, dim=1)
    valid_angle = angle > 0
    M = U.axis_angle_to_matrix(angle.unsqueeze(1) * axis / norms.unsqueeze(1))
    R = R_view.clone()
    R[valid_angle] = torch.bmm(M[valid_angle], R_view[valid_angle])
    return R


def R_to_allocentric(K, R, u, v):
    """math_util.py:595-623 (tensor branch): the inverse view rotation, M^T @ R where the viewing-ray angle is > 0."""
    fx, fy, sx, sy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    oray = torch.stack(((u - sx) / fx, (v - sy) / fy, torch.ones_like(u))).T
    oray = oray / torch.linalg.norm(oray, dim=1).unsqueeze(1)
    angle = torch.acos(oray[:, -1])
    axis = torch.zeros_like(oray)
    axis[:, 0] = axis[:, 0] - oray[:, 1]
    axis[:, 1] = axis[:, 1] + oray[:, 0]
    norms = torch.linalg.norm(axis, dim=1)
    valid_angle = angle > 0
    M = U.axis_angle_to_matrix(angle.unsqueeze(1) * axis / norms.unsqueeze(1))
    R_view = R.clone()
    R_view[valid_angle] = torch.bmm(M[valid_angle].transpose(2, 1), R[valid_angle])
    return R_view


def chamfer_loss(vals, target):
    """roi_heads.py:298-304"""
    B = vals.shape[0]
    l1 = (vals.view(B, 8, 1, 3) - target.view(B, 1, 8, 3)).abs().sum(-1)
    return l1.min(1).values.mean(-1) + l1.min(2).values.mean(-1)


def safely_reduce(loss):
    """roi_heads.py:932-940"""
    valid = (~loss.isinf()) & (~loss.isnan())
    return loss[valid].mean() if valid.any() else loss.mean() * 0.0


def cube_head_outputs_to_fused(xy, z, dims, pose6, uncert):
    """(n,K,2),(n,K,1),(n,K,3),(n,K,6),(n,K) -> (n, 13K) in the product's fused column order."""
    n = xy.shape[0]
    return torch.cat([xy.reshape(n, -1), z.reshape(n, -1), dims.reshape(n, -1), pose6.reshape(n, -1), uncert.reshape(n, -1)], 1)


E_CONSTANT = 2.71828183        # roi_heads.py:28
POSE_WIDTH = {"6d": 6, "quaternion": 4, "euler": 3}


def cube_losses(head, num_classes, boxes, classes, Ks_scaled, virtual_to_real, prior_mean, gt_boxes3D, gt_poses, *,
                prior_std=None, z_type="direct", dims_priors_enabled=True, dims_priors_func="exp", pose_type="6d",
                allocentric_pose=True, virtual_depth=True, chamfer_pose=True, inverse_z_weight=False, use_confidence=True,
                joint=True, loss_w=(1.0, 1.0, 1.0, 1.0, 1.0), disentangled=True, cluster_bins=1, z_scales=None, z_stats=None):
    """CubeHead.forward tail (cube_head.py:163,175-197) + ROIHeads3D._forward_cube training path (roi_heads.py:410-768) with
    DISENTANGLED_LOSS (every released config).  Defaults = configs/Base.yaml (z 'direct', 6d pose, dims priors 'exp', virtual
    depth, allocentric, chamfer + joint, confidence); the keyword switches follow MODEL.ROI_CUBE_HEAD.{Z_TYPE, DIMS_PRIORS_*,
    POSE_TYPE, ALLOCENTRIC_POSE, VIRTUAL_DEPTH, CHAMFER_POSE, INVERSE_Z_WEIGHT, USE_CONFIDENCE, LOSS_W_JOINT > 0, DISENTANGLED_LOSS,
    CLUSTER_BINS}; z_scales (K, bins) / z_stats (K, bins, 2) = ROIHeads3D.priors_z_scales / priors_z_stats (roi_heads.py:123-143).
    head (n, width*K) raw fused linear outputs [xy 2K | z K*bins (bin-major, cube_head.py:191-192) | dims 3K | pose Pn*K |
    uncert K if confidence];
    boxes (n,4) proposal boxes; classes (n,); Ks_scaled (n,3,3); virtual_to_real (n,); prior_mean / prior_std (n,3);
    gt_boxes3D (n,9); gt_poses (n,3,3); loss_w = (dims, pose, xy, z, joint) for the logged total.
    -> (losses dict (unweighted means), stats dict, extras)."""
    K = num_classes
    n = head.shape[0]
    ar = torch.arange(n)
    Pn = POSE_WIDTH[pose_type]
    nb = max(int(cluster_bins), 1)
    o_d, o_p, o_u = (2 + nb) * K, (5 + nb) * K, (5 + nb + Pn) * K
    box_2d_deltas = head[:, : 2 * K].view(n, K, 2)
    box_z = head[:, 2 * K: o_d].view(n, nb, K, 1) if nb > 1 else head[:, 2 * K: o_d].view(n, K, 1)
    box_dims = head[:, o_d: o_p].view(n, K, 3)
    raw_pose = head[:, o_p: o_u]
    if pose_type == "6d":                                                     # cube_head.py:175-176
        box_pose = U.rotation_6d_to_matrix(raw_pose.reshape(-1, 6))
    elif pose_type == "quaternion":                                           # cube_head.py:178-182
        quats = raw_pose.reshape(-1, 4)
        quats = quats / U._copysign(torch.sqrt((quats * quats).sum(1)), quats[:, 0])[:, None]
        box_pose = U.quaternion_to_matrix(quats)
    else:                                                                     # cube_head.py:184-185
        box_pose = U.euler_angles_to_matrix(raw_pose.reshape(-1, 3), "XYZ")
    box_pose = box_pose.view(n, K, 3, 3)
    src_w = boxes[:, 2] - boxes[:, 0]
    src_h = boxes[:, 3] - boxes[:, 1]
    if nb > 1:                                                                # roi_heads.py:432-442: nearest 2D-scale cluster
        src_scales = (src_h ** 2 + src_w ** 2).sqrt()
        assignments = (z_scales.detach().T.unsqueeze(0) - src_scales.unsqueeze(1).unsqueeze(2)).abs().argmin(1)      # (n, K)
        cube_z = box_z[ar, :, classes, :][ar, assignments[ar, classes]]
    else:
        cube_z = box_z[ar, classes, :]
    cube_dims = box_dims[ar, classes, :]
    cube_pose = box_pose[ar, classes, :, :]
    cube_uncert = head[:, o_u: o_u + K].clip(0.01)[ar, classes] if use_confidence else None
    cube_2d_deltas = box_2d_deltas[ar, classes, :]
    src_cx = boxes[:, 0] + 0.5 * src_w
    src_cy = boxes[:, 1] + 0.5 * src_h
    cube_x = src_cx + src_w * cube_2d_deltas[:, 0]
    cube_y = src_cy + src_h * cube_2d_deltas[:, 1]
    cube_xy = torch.stack((cube_x, cube_y), dim=1)
    cube_dims_norm = cube_dims
    if dims_priors_enabled:                                                   # roi_heads.py:467-484
        if dims_priors_func == "sigmoid":
            mn, mx = (prior_mean - 3 * prior_std).clip(0.0), prior_mean + 3 * prior_std
            cube_dims = mn + (mx - mn) * torch.sigmoid(cube_dims)             # util.scaled_sigmoid (math_util.py:969-978)
        else:
            cube_dims = torch.exp(cube_dims.clip(max=5)) * prior_mean
    else:
        cube_dims = torch.exp(cube_dims.clip(max=5))
    cube_pose_allocentric = cube_pose
    if allocentric_pose:                                                      # roi_heads.py:486-490
        cube_pose = R_from_allocentric(Ks_scaled, cube_pose, u=cube_x.detach(), v=cube_y.detach())
    cube_z = cube_z.squeeze(1)
    cube_z_norm = cube_z
    if z_type == "sigmoid":                                                   # roi_heads.py:493-500
        cube_z_norm = torch.sigmoid(cube_z)
        cube_z = cube_z_norm * 100
    elif z_type == "log":
        cube_z = torch.exp(cube_z)
    elif z_type == "clusters":                                                # roi_heads.py:501-522
        z_means = torch.gather(z_stats[:, :, 0].T.unsqueeze(0).repeat([n, 1, 1]), 1, assignments.unsqueeze(1)).squeeze(1).detach()
        z_stds = torch.gather(z_stats[:, :, 1].T.unsqueeze(0).repeat([n, 1, 1]), 1, assignments.unsqueeze(1)).squeeze(1).detach()
        z_means, z_stds = z_means[ar, classes], z_stds[ar, classes]
        z_mins, z_maxs = (z_means - 3 * z_stds).clip(0), z_means + 3 * z_stds
        cube_z = z_mins + (z_maxs - z_mins) * torch.sigmoid(cube_z)           # util.scaled_sigmoid (math_util.py:969-978)
    real_to_virtual = 1.0
    if virtual_depth:                                                         # roi_heads.py:398-407, 524-525
        cube_z = cube_z * virtual_to_real
        real_to_virtual = 1 / virtual_to_real
    fx, fy, sx, sy = Ks_scaled[:, 0, 0], Ks_scaled[:, 1, 1], Ks_scaled[:, 0, 2], Ks_scaled[:, 1, 2]
    gt_2d, gt_z, gt_dims = gt_boxes3D[:, :2], gt_boxes3D[:, 2], gt_boxes3D[:, 3:6]
    gt_x3d = gt_z * (gt_2d[:, 0] - sx) / fx
    gt_y3d = gt_z * (gt_2d[:, 1] - sy) / fy
    gt_3d = torch.stack((gt_x3d, gt_y3d, gt_z)).T
    gt_box3d = torch.cat((gt_3d, gt_dims), dim=1)
    gt_corners = get_cuboid_verts(gt_box3d, gt_poses)
    l1 = lambda a, b: F.smooth_l1_loss(a, b, reduction="none", beta=0.0).contiguous().view(n, -1).mean(dim=1)  # noqa: E731
    if disentangled:                                                          # roi_heads.py:567-603
        dis_z = torch.cat((torch.stack((cube_z * (gt_2d[:, 0] - sx) / fx, cube_z * (gt_2d[:, 1] - sy) / fy, cube_z)).T, gt_dims), dim=1)
        loss_z = l1(get_cuboid_verts(dis_z, gt_poses), gt_corners)
        dis_xy = torch.cat((torch.stack((gt_z * (cube_x - sx) / fx, gt_z * (cube_y - sy) / fy, gt_z)).T, gt_dims), dim=1)
        loss_xy = l1(get_cuboid_verts(dis_xy, gt_poses), gt_corners)
        pose_corners = get_cuboid_verts(gt_box3d, cube_pose)
        loss_pose = chamfer_loss(pose_corners, gt_corners) if chamfer_pose else l1(pose_corners, gt_corners)      # roi_heads.py:597-601
        loss_dims = l1(get_cuboid_verts(torch.cat((gt_3d, cube_dims), dim=1), gt_poses), gt_corners)
    else:                                                                     # roi_heads.py:606-649: losses in the network's output spaces
        e1 = lambda a, b: F.smooth_l1_loss(a, b, reduction="none", beta=0.0)  # noqa: E731
        gt_deltas = (gt_2d.clone() - torch.stack((src_cx, src_cy), dim=1)) / torch.stack((src_w, src_h), dim=1)
        loss_xy = e1(cube_2d_deltas, gt_deltas).mean(1)
        # with DIMS_PRIORS_ENABLED the reference divides (n,3) by the (n,2,3) priors (:620-622), which does not evaluate
        assert not dims_priors_enabled, "roi_heads.py:620-622 cannot be evaluated (shape mismatch in the reference)"
        loss_dims = e1(cube_dims_norm, torch.log(gt_dims)).mean(1)
        if allocentric_pose:
            gt_alloc = R_to_allocentric(Ks_scaled, gt_poses, u=cube_x.detach(), v=cube_y.detach())
            loss_pose = 1 - U.so3_relative_angle(cube_pose_allocentric, gt_alloc, eps=0.1, cos_angle=True)
        else:
            loss_pose = 1 - U.so3_relative_angle(cube_pose, gt_poses, eps=0.1, cos_angle=True)
        if z_type == "direct":
            loss_z = e1(cube_z, gt_z)
        elif z_type == "sigmoid":
            loss_z = e1(cube_z_norm, (gt_z * real_to_virtual / 100).clip(0, 1))
        elif z_type == "log":
            loss_z = e1(cube_z_norm, torch.log((gt_z * real_to_virtual).clip(0.01)))
        else:
            loss_z = e1(cube_z_norm, ((gt_z * real_to_virtual) - z_means) / z_stds)
    wd, wp, wxy, wz, wj = loss_w
    total = (loss_dims * wd + loss_pose * wp + loss_xy * wxy + loss_z * wz).detach()                            # roi_heads.py:651-662
    if joint:                                                                 # roi_heads.py:664-683
        jb = torch.cat((torch.stack((cube_z * (cube_x - sx) / fx, cube_z * (cube_y - sy) / fy, cube_z)).T, cube_dims), dim=1)
        jc = get_cuboid_verts(jb, cube_pose)
        loss_joint = chamfer_loss(jc, gt_corners) if (chamfer_pose and disentangled) else l1(jc, gt_corners)      # roi_heads.py:676-680
        valid_joint = loss_joint < float("inf")
        total = total + (loss_joint * wj).detach()
    z_error = (cube_z - gt_z).detach().abs()
    stats = {"Cube/z_error": z_error.mean().item(), "Cube/dims_error": (cube_dims - gt_dims).detach().abs().mean().item(),
             "Cube/xy_error": (cube_xy - gt_2d).detach().abs().mean().item(), "Cube/z_close": (z_error < 0.20).float().mean().item(),
             "Cube/total_3D_loss": safely_reduce(total).item()}
    sf = torch.ones_like(loss_dims)
    if inverse_z_weight:                                                      # roi_heads.py:697-719
        sf = sf * (1 / torch.log(gt_z.clip(E_CONSTANT)))
    losses = {}
    if use_confidence:                                                        # roi_heads.py:721-740
        sf = sf * (SQRT_2 * torch.exp(-cube_uncert))
        losses["Cube/uncert"] = safely_reduce(cube_uncert.clone())
        stats["Cube/conf"] = torch.exp(-cube_uncert).mean().item()
    losses.update({"Cube/loss_dims": safely_reduce(loss_dims * sf), "Cube/loss_xy": safely_reduce(loss_xy * sf),
                   "Cube/loss_z": safely_reduce(loss_z * sf), "Cube/loss_pose": safely_reduce(loss_pose * sf)})
    if joint and valid_joint.any():
        losses["Cube/loss_joint"] = safely_reduce((loss_joint * sf)[valid_joint])
    extras = {"cube_x": cube_x, "cube_y": cube_y, "cube_z": cube_z, "cube_dims": cube_dims, "cube_pose": cube_pose,
              "cube_uncert": cube_uncert}
    return losses, stats, extras
