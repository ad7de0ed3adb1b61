"""oracle/model_oracle.py -- TEST INFRASTRUCTURE (CPU oracle of the WHOLE training step).

A plain-PyTorch (CPU, fp32, autograd) model with the reference's parameter names that composes
  * the DLA-34 bottom-up restated from /root/reference/cubercnn/modeling/backbone/dla.py:40-68,156-297,463-482
  * the upstream pieces of oracle/upstream.py (FPN, StandardRPNHead, anchors, ROIPooler, proposals)
  * the reference's own losses / sampling restated in oracle/cubercnn_oracle.py
into `RCNN3D.forward` (rcnn3d.py:41-77).  It travels to the GPU box, where it is (a) the oracle of
the full-size parity test and (b) the `cpu_baseline` leg of bench.py.  In the build container it is
pinned against the reference itself through the golden fixtures (tests/test_model_parity.py)."""
import math
import time

import torch
import torch.nn.functional as F
from torch import nn

from oracle import cubercnn_oracle as O
from oracle import upstream as U
from omni3d_amd.d2.structures import Boxes, ImageList


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x, residual=None):
        residual = x if residual is None else residual
        out = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(out)) + residual)


class Bottleneck(nn.Module):
    """dla.py:71-109 (expansion 2)"""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        mid = cout // 2
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)

    def forward(self, x, residual=None):
        residual = x if residual is None else residual
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        return F.relu(self.bn3(self.conv3(out)) + residual)


def bottleneck_x(cardinality):
    """dla.py:112-153 (cardinality 32; dla102x2 sets 64)"""

    class BottleneckX(nn.Module):
        def __init__(self, cin, cout, stride=1):
            super().__init__()
            mid = cout * cardinality // 32
            self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(mid)
            self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False, groups=cardinality)
            self.bn2 = nn.BatchNorm2d(mid)
            self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(cout)

        def forward(self, x, residual=None):
            residual = x if residual is None else residual
            out = F.relu(self.bn1(self.conv1(x)))
            out = F.relu(self.bn2(self.conv2(out)))
            return F.relu(self.bn3(self.conv3(out)) + residual)
    return BottleneckX


class Root(nn.Module):
    def __init__(self, cin, cout, residual=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.residual = residual

    def forward(self, *x):
        y = self.bn(self.conv(torch.cat(x, 1)))
        return F.relu(y + x[0] if self.residual else y)


class Tree(nn.Module):
    def __init__(self, levels, cin, cout, stride=1, level_root=False, root_dim=0, block=BasicBlock, root_residual=False):
        super().__init__()
        root_dim = 2 * cout if root_dim == 0 else root_dim
        if level_root:
            root_dim += cin
        if levels == 1:
            self.tree1, self.tree2 = block(cin, cout, stride), block(cout, cout, 1)
            self.root = Root(root_dim, cout, root_residual)
        else:
            self.tree1 = Tree(levels - 1, cin, cout, stride, root_dim=0, block=block, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, cout, cout, root_dim=root_dim + cout, block=block, root_residual=root_residual)
        self.level_root, self.levels, self.stride = level_root, levels, stride
        self.project = nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout)) if cin != cout else None

    def forward(self, x, residual=None, children=None):
        children = [] if children is None else children
        bottom = F.max_pool2d(x, self.stride, self.stride) if self.stride > 1 else x
        residual = self.project(bottom) if self.project is not None else bottom
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1(x, residual)
        if self.levels == 1:
            return self.root(self.tree2(x1), x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


# MODEL.DLA.TYPE -> (levels, channels, block, residual_root)   (dla.py:312-414; dla102x2 = BottleneckX with cardinality 64)
DLA_VARIANTS = {
    "dla34": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], BasicBlock, False),
    "dla46_c": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 64, 128, 256], Bottleneck, False),
    "dla60": ([1, 1, 1, 2, 3, 1], [16, 32, 128, 256, 512, 1024], Bottleneck, False),
    "dla102": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], Bottleneck, True),
    "dla169": ([1, 1, 2, 3, 5, 1], [16, 32, 128, 256, 512, 1024], Bottleneck, True),
    "dla46x_c": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 64, 128, 256], bottleneck_x(32), False),
    "dla60x_c": ([1, 1, 1, 2, 3, 1], [16, 32, 64, 64, 128, 256], bottleneck_x(32), False),
    "dla60x": ([1, 1, 1, 2, 3, 1], [16, 32, 128, 256, 512, 1024], bottleneck_x(32), False),
    "dla102x": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], bottleneck_x(32), True),
    "dla102x2": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], bottleneck_x(64), True),
}


class DLA34(U.Backbone):
    """DLABackbone (dla.py:417-482) of any non-grouped MODEL.DLA.TYPE; the default is the dla34 of BASELINE.json"""

    def __init__(self, variant="dla34"):
        super().__init__()
        lv, c, block, rr = DLA_VARIANTS[variant]

        def cbr(cin, cout, k, s, p):
            return nn.Sequential(nn.Conv2d(cin, cout, k, s, p, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.base_layer, self.level0, self.level1 = cbr(3, c[0], 7, 1, 3), cbr(c[0], c[0], 3, 1, 1), cbr(c[0], c[1], 3, 2, 1)
        self.level2 = Tree(lv[2], c[1], c[2], 2, level_root=False, block=block, root_residual=rr)
        self.level3 = Tree(lv[3], c[2], c[3], 2, level_root=True, block=block, root_residual=rr)
        self.level4 = Tree(lv[4], c[3], c[4], 2, level_root=True, block=block, root_residual=rr)
        self.level5 = Tree(lv[5], c[4], c[5], 2, level_root=True, block=block, root_residual=rr)
        self._out_feature_channels = {"p2": c[2], "p3": c[3], "p4": c[4], "p5": c[5], "p6": c[5]}
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    def forward(self, x):
        x = self.level1(self.level0(self.base_layer(x)))
        p2 = self.level2(x)
        p3 = self.level3(p2)
        p4 = self.level4(p3)
        p5 = self.level5(p4)
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": F.max_pool2d(p5, kernel_size=1, stride=2, padding=0)}


class ResNet34(U.Backbone):
    """cubercnn/modeling/backbone/resnet.py:12-65 over the restated torchvision resnet34 (oracle/upstream.py)."""

    def __init__(self, depth=34):
        super().__init__()
        base = {18: U.tv_resnet18, 34: U.tv_resnet34, 50: U.tv_resnet50, 101: U.tv_resnet101}[depth](False)
        for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"):
            setattr(self, name, getattr(base, name))
        e = base.block.expansion
        self._out_feature_channels = {"p2": 64 * e, "p3": 128 * e, "p4": 256 * e, "p5": 512 * e, "p6": 512 * e}
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        p2 = self.layer1(x)
        p3 = self.layer2(p2)
        p4 = self.layer3(p3)
        p5 = self.layer4(p4)
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": F.max_pool2d(p5, kernel_size=1, stride=2, padding=0)}


class _Seq(nn.Module):
    pass


class ModelOracle(nn.Module):
    """State-dict compatible with the reference's RCNN3D (cubercnn_DLA34_FPN)."""

    def __init__(self, priors, num_classes=50, rpn_batch=256, roi_batch=512, pre_nms=2000, post_nms=1000, backbone="dla34"):
        super().__init__()
        self.K, self.rpn_batch, self.roi_batch, self.pre_nms, self.post_nms = num_classes, rpn_batch, roi_batch, pre_nms, post_nms
        names = ["p2", "p3", "p4", "p5", "p6"]
        self.backbone = (U.FPN(ResNet34(), names, 256, top_block=U.LastLevelMaxPool()) if backbone == "resnet34"
                         else U.FPN(DLA34(), names, 256))
        self.proposal_generator = _Seq()
        self.proposal_generator.rpn_head = U.StandardRPNHead(in_channels=256, num_anchors=3, box_dim=4)
        self.anchor_gen = U.DefaultAnchorGenerator(sizes=[[32], [64], [128], [256], [512]], aspect_ratios=[[0.5, 1.0, 2.0]],
                                                   strides=[4, 8, 16, 32, 64], offset=0.0)
        rh = self.roi_heads = _Seq()
        rh.box_head = _Seq()
        rh.box_head.fc1, rh.box_head.fc2 = nn.Linear(12544, 1024), nn.Linear(1024, 1024)
        rh.box_predictor = _Seq()
        rh.box_predictor.cls_score, rh.box_predictor.bbox_pred = nn.Linear(1024, num_classes + 1), nn.Linear(1024, 4 * num_classes)
        ch = rh.cube_head = _Seq()
        ch.feature_generator = _Seq()
        ch.feature_generator.fc1, ch.feature_generator.fc2 = nn.Linear(12544, 1024), nn.Linear(1024, 1024)
        ch.bbox_3D_dims, ch.bbox_3D_center_deltas = nn.Linear(1024, 3 * num_classes), nn.Linear(1024, 2 * num_classes)
        ch.bbox_3D_pose, ch.bbox_3D_center_depth = nn.Linear(1024, 6 * num_classes), nn.Linear(1024, num_classes)
        ch.bbox_3D_uncertainty = nn.Linear(1024, num_classes)
        rh.priors_dims_per_cat = nn.Parameter(torch.FloatTensor(priors["priors_dims_per_cat"]).unsqueeze(0))
        rh.priors_z_scales = nn.Parameter(torch.ones(num_classes, 1))
        self.pooler = U.ROIPooler(7, (1 / 4, 1 / 8, 1 / 16, 1 / 32, 1 / 64), 0, "ROIAlignV2")
        self.register_buffer("pixel_mean", torch.tensor([103.530, 116.280, 123.675]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor([57.375, 57.120, 58.395]).view(-1, 1, 1), False)
        self.b2b = U.Box2BoxTransform((1.0, 1.0, 1.0, 1.0))

    def forward(self, batch, E_rpn, E_roi, virtual_focal=512.0, proposals=None):
        """proposals: optional list of (P_n, 4) boxes that REPLACE the first stage's own proposal list in the second stage
        (stage-wise comparison: near-tied objectness scores can swap under any change of fp32 summation order or precision,
        so a second stage is only comparable across implementations / precisions on one fixed proposal list)."""
        B = len(batch)
        images = ImageList.from_tensors([(x["image"].float() - self.pixel_mean) / self.pixel_std for x in batch], 64)
        feats = self.backbone(images.tensor)
        fl = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        anchors = torch.cat([a.tensor for a in self.anchor_gen(fl)])
        lg, dl = self.proposal_generator.rpn_head(fl)
        logits = torch.cat([s.permute(0, 2, 3, 1).flatten(1) for s in lg], 1)
        deltas = torch.cat([x.view(B, -1, 4, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2) for x in dl], 1)
        gts = [x["instances"] for x in batch]
        labels, mgt = [], []
        for n, g in enumerate(gts):
            ok = g.gt_classes >= 0
            lab, midx, _, _ = O.rpn_label_and_sample(anchors, g.gt_boxes.tensor[ok], g.gt_boxes.tensor[~ok], E_rpn[n],
                                                     batch_size_per_image=self.rpn_batch)
            labels.append(lab)
            mgt.append(g.gt_boxes.tensor[ok][midx])
        losses, _ = O.rpn_losses_iouness(anchors, logits, deltas, torch.stack(labels), torch.stack(mgt), self.rpn_batch)
        # proposals (detectron2 predict_proposals)
        with torch.no_grad():
            per_level, off = [], 0
            for s in lg:
                n = s.shape[1] * s.shape[2] * s.shape[3]
                per_level.append((off, n))
                off += n
            dec = self.b2b.apply_deltas(deltas.detach().reshape(-1, 4), anchors.unsqueeze(0).expand(B, -1, -1).reshape(-1, 4)).view(B, -1, 4)
            props = U.find_top_rpn_proposals([dec[:, o:o + n] for o, n in per_level], [logits.detach()[:, o:o + n] for o, n in per_level],
                                             images.image_sizes, 0.7, self.pre_nms, self.post_nms, 0.0, True)
        self.last_proposals = [p.proposal_boxes.tensor.detach().clone() for p in props]
        if proposals is not None:
            from omni3d_amd.d2.structures import Instances
            inj = []
            for n, bx in enumerate(proposals):
                inst = Instances(images.image_sizes[n])
                inst.proposal_boxes = Boxes(bx.to(anchors.dtype))
                inj.append(inst)
            props = inj
        # ROI heads
        sb, sc, sgb, s3d, spose, simg = [], [], [], [], [], []
        for n, g in enumerate(gts):
            ok = g.gt_classes >= 0
            bx, cls, midx, _ = O.roi_label_and_sample(props[n].proposal_boxes.tensor, g.gt_boxes.tensor[ok], g.gt_classes[ok],
                                                      g.gt_boxes.tensor[~ok], E_roi[n], num_classes=self.K, batch_size_per_image=self.roi_batch)
            sb.append(bx); sc.append(cls); sgb.append(g.gt_boxes.tensor[ok][midx])
            s3d.append(g.gt_boxes3D[ok][midx]); spose.append(g.gt_poses[ok][midx])
        rh = self.roi_heads
        x = self.pooler(fl, [Boxes(b) for b in sb]).flatten(1)
        x = F.relu(rh.box_head.fc2(F.relu(rh.box_head.fc1(x))))
        cls_all, box_all = torch.cat(sc), torch.cat(sb)
        losses.update(O.fast_rcnn_losses(rh.box_predictor.cls_score(x), rh.box_predictor.bbox_pred(x), cls_all, box_all, torch.cat(sgb), self.K))
        fg = [(c >= 0) & (c < self.K) for c in sc]
        fb = [b[m] for b, m in zip(sb, fg)]
        nfg = [int(m.sum()) for m in fg]
        if sum(nfg) > 0:
            xc = self.pooler(fl, [Boxes(b) for b in fb]).flatten(1)
            ch = rh.cube_head
            f = F.relu(ch.feature_generator.fc2(F.relu(ch.feature_generator.fc1(xc))))
            head = torch.cat([ch.bbox_3D_center_deltas(f), ch.bbox_3D_center_depth(f), ch.bbox_3D_dims(f), ch.bbox_3D_pose(f),
                              ch.bbox_3D_uncertainty(f)], 1)
            Km, v2r = [], []
            for n, (info, k) in enumerate(zip(batch, nfg)):
                h_net = images.image_sizes[n][0]
                r = info["height"] / h_net
                Kt = torch.tensor(info["K"], dtype=anchors.dtype) / r
                Kt[2, 2] = 1
                Km.append(Kt.unsqueeze(0).repeat(k, 1, 1))
                v2r.append(torch.full((k,), (h_net * info["K"][1][1]) / (virtual_focal * (h_net * r)), dtype=anchors.dtype))
            fcls = torch.cat([c[m] for c, m in zip(sc, fg)])
            prior_mean = rh.priors_dims_per_cat.detach()[0, fcls, 0, :]
            cl, _, _ = O.cube_losses(head, self.K, torch.cat(fb), fcls, torch.cat(Km), torch.cat(v2r), prior_mean,
                                     torch.cat([g[m] for g, m in zip(s3d, fg)]), torch.cat([p[m] for p, m in zip(spose, fg)]))
            losses.update(cl)
        self.last_labels = torch.stack(labels)
        self.last_roi_boxes = [b.detach().clone() for b in sb]       # sampled ROI boxes per image (tests: set comparison)
        return losses


def to_double(batch):
    """deep copy of a batch with every floating-point ground-truth tensor in float64 (for the fp64 run of the oracle)"""
    import copy
    out = copy.deepcopy(batch)
    for b in out:
        inst = b.get("instances")
        if inst is None:
            continue
        for k, v in list(inst._fields.items()):
            if torch.is_tensor(v) and v.is_floating_point():
                inst._fields[k] = v.double()
            elif hasattr(v, "tensor"):
                v.tensor = v.tensor.double()
    return out


def run_fp64(priors, state_dict, batch, E_rpn, E_roi, proposals, backbone="dla34", **kw):
    """The same oracle evaluated in float64 on the same weights / variates / proposal list: the yardstick that separates
    fp32 rounding (of the CPU oracle AND of the HIP path) from real differences.  -> (losses, {name: grad}, oracle)"""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        o = ModelOracle(priors, backbone=backbone, **kw)
        o.load_state_dict(state_dict, strict=True)
        o = o.double()
        o.train()
        losses = o(to_double(batch), E_rpn.double(), E_roi.double(), proposals=[p.double() for p in proposals])
        sum(losses.values()).backward()
    finally:
        torch.set_default_dtype(prev)
    return ({k: float(v.detach()) for k, v in losses.items()}, {n: p.grad for n, p in o.named_parameters() if p.grad is not None}, o)


def time_training(priors, batches=(2, 4), size=512, warmup=3, timed=10, budget_s=150.0):
    """bench.py cpu_baseline leg: forward + losses + backward + SGD of the CPU oracle on the host cores, batch 2
    and batch 4, >= 3 warm-up + >= 10 timed iterations each, median.  `budget_s` bounds the whole leg: on a slow host the timed
    count of the later batch shrinks (never below 5) and the sample string says what was run."""
    import os
    from omni3d_amd import synthetic
    cores = min(os.cpu_count() or 1, 32)   # more intra-op threads than this only adds contention on big hosts
    try:    # (the container's CFS quota, not the host's core count, is what the threads can really use: 16 CPUs on the GPU boxes)
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(cores)
    t_leg = time.perf_counter()
    per_batch = {}
    for bi, images in enumerate(batches):
        torch.manual_seed(0)
        model = ModelOracle(priors)
        model.train()
        # the step's cost does not depend on the learning rate; a random-init Cube R-CNN at the reference's 0.02 * batch / 32 can
        # diverge inside 13 iterations (NaN proposals raise in the restated find_top_rpn_proposals), so the timing run crawls
        opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4)
        batch = synthetic.make_batch(images, size, size, num_gt=8, seed=1000, priors=priors)
        A = 3 * sum((size // s) ** 2 for s in (4, 8, 16, 32, 64))
        g = torch.Generator().manual_seed(1)
        share = budget_s * (bi + 1) / len(batches)          # cumulative share of the budget this batch may run into
        times = []
        for it in range(warmup + timed):
            done = len(times) - warmup
            if done >= 5 and time.perf_counter() - t_leg > share:
                break
            E_rpn = torch.empty(images, A).exponential_(generator=g)
            E_roi = torch.empty(images, 2048).exponential_(generator=g)
            t0 = time.perf_counter()
            opt.zero_grad()
            losses = model(batch, E_rpn, E_roi)
            sum(losses.values()).backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        kept = sorted(times[warmup:])
        med = kept[len(kept) // 2]
        per_batch[images] = {"images_per_s": images / med, "median_s_per_iter": med, "warmup": warmup, "timed": len(kept)}
    head = per_batch[batches[-1]]          # the batch BASELINE's metric is quoted on (4 / GPU)
    try:
        from omni3d_amd.profile_io import host_cores
        host = host_cores()
    except Exception:  # noqa: BLE001
        host = None
    return {"value": head["images_per_s"], "unit": "images/s", "cores": cores, "kind": "port", "host": host,
            "by_batch": {str(k): v for k, v in per_batch.items()},
            "sample": f"oracle/model_oracle.py (plain-PyTorch CPU port of the reference path, pinned to fixtures written by the reference's own "
                      f"files), {size}x{size}, fwd+10 losses+bwd+SGD; `value` = batch {batches[-1]}: median of {head['timed']} timed iterations "
                      f"after {warmup} warm-up; batch {batches[0]} in by_batch; torch threads={cores}, os.cpu_count()={os.cpu_count()}, "
                      f"{time.perf_counter() - t_leg:.0f} s of CPU work"}
