"""RPN / ROI / ROIAlign / loss / cube / optimizer kernels vs the CPU oracle
(oracle/upstream.py + oracle/cubercnn_oracle.py).  Integer outputs are compared exactly."""
import numpy as np
import pytest
import torch

from oracle import cubercnn_oracle as O
from oracle import upstream as U
from omni3d_amd.d2.structures import Boxes


def _anchors(sizes, strides, hw):
    gen = U.DefaultAnchorGenerator(sizes=[[s] for s in sizes], aspect_ratios=[[0.5, 1.0, 2.0]], strides=strides, offset=0.0)
    feats = [torch.zeros(1, 1, h, w) for h, w in hw]
    return torch.cat([b.tensor for b in gen(feats)])


def _rand_boxes(g, n, W, H, smin=8, smax=80):
    cx, cy = torch.rand(n, generator=g) * W, torch.rand(n, generator=g) * H
    w, h = smin + torch.rand(n, generator=g) * (smax - smin), smin + torch.rand(n, generator=g) * (smax - smin)
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, W)
    b[:, 1::2] = b[:, 1::2].clamp(0, H)
    return b


def _rpn_setup(dev, seed=0, crowded=0):
    g = torch.Generator().manual_seed(seed)
    hw, strides, sizes = [(16, 16), (8, 8), (4, 4)], [4, 8, 16], [16, 32, 64]
    anchors = _anchors(sizes, strides, hw)
    B = 2
    # crowded: one image with more ground-truth boxes than the kernels' LDS staging held before round 3 (256)
    gts = [_rand_boxes(g, crowded or 5, 64, 64), _rand_boxes(g, 3, 64, 64)]
    igns = [_rand_boxes(g, 1, 64, 64, 30, 60), torch.zeros(0, 4)]
    E = torch.empty(B, anchors.shape[0]).exponential_(generator=g)
    return anchors, gts, igns, E, hw


def _offsets(lst):
    return torch.tensor(np.concatenate([[0], np.cumsum([len(x) for x in lst])]), dtype=torch.int32)


def _run_rpn_labels(dev, batch_per_image, pos_frac, seed, crowded=0):
    from omni3d_amd.kernels import det, select
    anchors, gts, igns, E, hw = _rpn_setup(dev, seed, crowded)
    A, B = anchors.shape[0], len(gts)
    gt, ign = torch.cat(gts), torch.cat(igns)
    gt_off, ign_off = _offsets(gts), _offsets(igns)
    m = det.rpn_match(anchors.to(dev), gt.to(dev), gt_off.to(dev), E.to(dev))
    kpos = int(batch_per_image * pos_frac)
    pv, pi = select.topk_rows(m["key_pos"], max(kpos, 1))
    nv, ni = select.topk_rows(m["key_neg"], batch_per_image)
    labels, counts = det.rpn_finalize_labels(anchors.to(dev), gt_off.to(dev), ign.to(dev) if len(ign) else torch.zeros(1, 4).to(dev),
                                             ign_off.to(dev), m, pv, pi, nv, ni, batch_per_image, 0.5)
    for n in range(B):
        ref_labels, ref_midx, ref_miou, _ = O.rpn_label_and_sample(anchors, gts[n], igns[n], E[n], batch_size_per_image=batch_per_image,
                                                                  positive_fraction=pos_frac)
        assert torch.equal(labels[n].cpu(), ref_labels), n
        assert torch.equal(m["matched_idx"][n].cpu().long(), ref_midx)
        assert torch.equal(m["matched_val"][n].cpu(), ref_miou)
    return anchors, gts, labels, m, hw


def _run_rpn_loss(dev):
    from omni3d_amd.kernels import det
    anchors, gts, labels, m, hw = _run_rpn_labels(dev, 64, 1.0, 3)
    B, A = labels.shape
    g = torch.Generator().manual_seed(11)
    levels = [torch.randn(B, h, w, 16, generator=g) for h, w in hw]
    pack = det.LevelPack([t.to(dev) for t in levels])
    # oracle views: logits (B,A), deltas (B,A,4)
    logits = torch.cat([t[..., :3].reshape(B, -1) for t in levels], 1).requires_grad_(True)
    deltas = torch.cat([t[..., 3:15].reshape(B, -1, 4) for t in levels], 1).requires_grad_(True)
    assert torch.equal(det.rpn_gather_logits(pack).cpu(), logits.detach())
    gt = torch.cat(gts)
    gt_off = _offsets(gts)
    midx = m["matched_idx"].cpu().long()
    mgt = torch.stack([gts[n][midx[n]] for n in range(B)])
    ref, stats = O.rpn_losses_iouness(anchors, logits, deltas, labels.cpu(), mgt, batch_size_per_image=64)
    sums = det.rpn_loss_fwd(pack, anchors.to(dev), labels, m["matched_idx"], gt.to(dev), gt_off.to(dev)).cpu()
    norm = 64 * B
    assert abs(sums[0].item() / norm - ref["rpn/cls"].item()) < 1e-5
    assert abs(sums[1].item() / norm - ref["rpn/loc"].item()) < 1e-4
    assert sums[2].item() / B == stats["rpn/num_pos_anchors"] and sums[3].item() / B == stats["rpn/num_neg_anchors"]
    assert abs(sums[4].item() / sums[2].item() - stats["rpn/conf_pos_anchors"]) < 1e-5
    (2.0 * ref["rpn/cls"] + 0.5 * ref["rpn/loc"]).backward()
    grads = det.rpn_loss_bwd(pack, anchors.to(dev), labels, m["matched_idx"], gt.to(dev), gt_off.to(dev),
                             torch.tensor([2.0]).to(dev), torch.tensor([0.5]).to(dev), 1.0 / norm)
    dlog = torch.cat([t[..., :3].reshape(B, -1) for t in grads], 1).cpu()
    ddel = torch.cat([t[..., 3:15].reshape(B, -1, 4) for t in grads], 1).cpu()
    assert (dlog - logits.grad).abs().max() < 1e-6 and (ddel - deltas.grad).abs().max() < 1e-6
    assert all(t[..., 15].abs().max() == 0 for t in grads)
    # MODEL.RPN.OBJECTNESS_UNCERTAINTY 'none': 0 / 1 objectness targets over every sampled anchor, unweighted L1 (rpn.py:181-195)
    logits.grad = deltas.grad = None
    refp = O.rpn_losses_plain(anchors, logits, deltas, labels.cpu(), mgt, batch_size_per_image=64)
    sums = det.rpn_loss_fwd(pack, anchors.to(dev), labels, m["matched_idx"], gt.to(dev), gt_off.to(dev), plain=True).cpu()
    assert abs(sums[0].item() / norm - refp["rpn/cls"].item()) < 1e-5 * max(1.0, refp["rpn/cls"].item())
    assert abs(sums[1].item() / norm - refp["rpn/loc"].item()) < 1e-4 * max(1.0, refp["rpn/loc"].item())
    assert sums[2].item() / B == stats["rpn/num_pos_anchors"] and sums[3].item() / B == stats["rpn/num_neg_anchors"]
    (2.0 * refp["rpn/cls"] + 0.5 * refp["rpn/loc"]).backward()
    grads = det.rpn_loss_bwd(pack, anchors.to(dev), labels, m["matched_idx"], gt.to(dev), gt_off.to(dev),
                             torch.tensor([2.0]).to(dev), torch.tensor([0.5]).to(dev), 1.0 / norm, plain=True)
    dlog = torch.cat([t[..., :3].reshape(B, -1) for t in grads], 1).cpu()
    ddel = torch.cat([t[..., 3:15].reshape(B, -1, 4) for t in grads], 1).cpu()
    assert (dlog - logits.grad).abs().max() < 1e-6 and (ddel - deltas.grad).abs().max() < 1e-6
    assert all(t[..., 15].abs().max() == 0 for t in grads)
    # decode of selected anchors == detectron2 apply_deltas + clip + nonempty
    b2b = U.Box2BoxTransform((1.0, 1.0, 1.0, 1.0))
    a_off = np.concatenate([[0], np.cumsum([h * w * 3 for h, w in hw])])
    k = 20
    slot_level = torch.tensor(sum([[l] * k for l in range(len(hw))], []), dtype=torch.int32)
    idx = torch.stack([torch.cat([torch.randperm(hw[l][0] * hw[l][1] * 3, generator=g)[:k] for l in range(len(hw))]) for _ in range(B)]).int()
    idx[0, 3] = -1
    image_hw = torch.tensor([[64, 64], [60, 50]], dtype=torch.int32)
    boxes, valid = det.rpn_decode(pack, slot_level.to(dev), idx.to(dev), anchors.to(dev), image_hw.to(dev))
    for n in range(B):
        for j in range(len(slot_level)):
            if idx[n, j] < 0:
                assert valid[n, j] == 0
                continue
            a = int(a_off[slot_level[j]] + idx[n, j])
            pb = Boxes(b2b.apply_deltas(deltas[n, a].detach()[None], anchors[a][None]))
            pb.clip(tuple(image_hw[n].tolist()))
            # expf on the GPU and exp on the CPU differ in the last ulp: fp32 tolerance, not bit-exact
            assert (boxes[n, j].cpu() - pb.tensor[0]).abs().max() < 1e-4, (n, j)
            assert bool(valid[n, j]) == bool(pb.nonempty()[0])


def _run_roi_sample(dev, crowded=0):
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(5)
    B, pmax, Kc = 2, 300, 50
    n0 = crowded or 6          # crowded: more ground truth in one image than the kernels staged before round 3 (256), all appended
    gts = [_rand_boxes(g, n0, 200, 200, 20, 90), _rand_boxes(g, 2, 200, 200, 20, 90)]
    gcls = [torch.randint(0, Kc, (n0,), generator=g), torch.randint(0, Kc, (2,), generator=g)]
    igns = [_rand_boxes(g, 2, 200, 200, 60, 150), torch.zeros(0, 4)]
    props = torch.stack([_rand_boxes(g, pmax, 200, 200, 10, 100) for _ in range(B)])
    # make some proposals overlap GT strongly
    for n in range(B):
        for j in range(40):
            props[n, j] = gts[n][j % len(gts[n])] + torch.randn(4, generator=g) * 2.0
    pcount = torch.tensor([pmax, 250], dtype=torch.int32)
    E = torch.empty(B, det.ROI_MAXC).exponential_(generator=g)
    gt_off, ign_off = _offsets(gts), _offsets(igns)
    ob, oc, og, oi, cnt, orow, ofirst = det.roi_sample(props.to(dev), pcount.to(dev), torch.cat(gts).to(dev), torch.cat(gcls).int().to(dev),
                                         gt_off.to(dev), torch.cat(igns).to(dev), ign_off.to(dev), E.to(dev), 0.5, 0.5, Kc, 128, 0.25)
    for n in range(B):
        rb, rc, rm, rs = O.roi_label_and_sample(props[n, : pcount[n]], gts[n], gcls[n], igns[n], E[n], num_classes=Kc,
                                                batch_size_per_image=128, positive_fraction=0.25)
        ns = len(rs)
        assert int(cnt[n].sum()) == ns
        assert torch.equal(ob[n, :ns].cpu(), rb)
        assert torch.equal(oc[n, :ns].cpu().long(), rc)
        assert torch.equal(og[n, :ns].cpu().long() - int(gt_off[n]), rm)
        assert (oc[n, ns:] == -2).all()
    # round 6: the rows the loss kernels index with (`clamp(min=0)` of the matched rows) and the cube head's contiguous prefix
    assert torch.equal(orow.cpu(), og.cpu().clamp(min=0)) and ofirst == (None, None, None)
    out = det.roi_sample(props.to(dev), pcount.to(dev), torch.cat(gts).to(dev), torch.cat(gcls).int().to(dev), gt_off.to(dev),
                         torch.cat(igns).to(dev), ign_off.to(dev), E.to(dev), 0.5, 0.5, Kc, 128, 0.25, first=32)
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(out[:6], (ob, oc, og, oi, cnt, orow)))
    fb, fc, fr = out[6]
    assert torch.equal(fb.cpu(), ob[:, :32].cpu()) and torch.equal(fc.cpu(), oc[:, :32].cpu()) and torch.equal(fr.cpu(), orow[:, :32].cpu())
    # in-kernel draws (csrc/philox.h): E = None + a DrawState.  The sample is a valid one (same counts: they do not depend on the
    # variates), repeats exactly from the same (seed, counter), differs from the next draw, and the kernel advances the counter
    from omni3d_amd.kernels.glue import DrawState
    args = (props.to(dev), pcount.to(dev), torch.cat(gts).to(dev), torch.cat(gcls).int().to(dev), gt_off.to(dev), torch.cat(igns).to(dev),
            ign_off.to(dev), None, 0.5, 0.5, Kc, 128, 0.25)
    torch.manual_seed(5)
    d1 = DrawState()
    a1 = det.roi_sample(*args, draw=d1)
    a2 = det.roi_sample(*args, draw=d1)
    assert d1.state.cpu().tolist()[1] == 2 and int(d1.ticket.cpu()) == 0
    torch.manual_seed(5)
    d2 = DrawState()
    b1 = det.roi_sample(*args, draw=d2)
    assert torch.equal(a1[4].cpu(), cnt.cpu()) and torch.equal(a2[4].cpu(), cnt.cpu())
    assert torch.equal(a1[0].cpu(), b1[0].cpu()) and torch.equal(a1[1].cpu(), b1[1].cpu())          # same seed, same counter
    assert not torch.equal(a1[0].cpu(), a2[0].cpu())                                                   # next draw: another sample
    for n in range(B):      # every sampled box is one of the candidates, foreground slots carry foreground classes
        ns = int(a1[4][n].sum())
        cand = torch.cat([props[n, : pcount[n]], gts[n]])
        assert all((cand == a1[0][n, j].cpu()).all(1).any() for j in range(ns))
        assert (a1[1][n, : int(a1[4][n, 0])].cpu() < Kc).all() and (a1[1][n, int(a1[4][n, 0]): ns].cpu() == Kc).all()


def _run_philox(dev):
    """the variates themselves: Exp(1) -- mean 1, variance 1, the right tail -- through the RPN matcher's sampling keys"""
    from omni3d_amd.kernels import det
    from omni3d_amd.kernels.glue import DrawState
    g = torch.Generator().manual_seed(3)
    A = 60000
    anchors = _rand_boxes(g, A, 400, 400, 8, 64)
    gt = torch.tensor([[10.0, 10, 50, 60]])
    gt_off = torch.tensor([0, 1, 1], dtype=torch.int32)       # image 0: one box, image 1: none (every anchor is a negative)
    torch.manual_seed(11)
    d = DrawState()
    m = det.rpn_match(anchors.to(dev), gt.to(dev), gt_off.to(dev), None, draw=d, B=2)
    keys = m["key_neg"][1].cpu().double()                      # (0 + eps) / e for every anchor of the empty image
    e = 1e-4 / keys
    assert torch.isfinite(e).all() and (e > 0).all()
    assert abs(float(e.mean()) - 1.0) < 0.02 and abs(float(e.var()) - 1.0) < 0.05
    for q, want in ((1.0, 0.3679), (2.0, 0.1353), (4.0, 0.0183)):
        assert abs(float((e > q).double().mean()) - want) < 0.01
    m2 = det.rpn_match(anchors.to(dev), gt.to(dev), gt_off.to(dev), None, draw=d, B=2)
    e2 = 1e-4 / m2["key_neg"][1].cpu().double()
    assert abs(float(((e - 1) * (e2 - 1)).mean())) < 0.02     # consecutive draws are uncorrelated
    assert d.state.cpu().tolist()[1] == 2 and int(d.ticket.cpu()) == 0
    # the two images of one draw are different streams
    lab = m["match_label"][0].cpu()
    e0 = (m["matched_val"][0].cpu().double() + 1e-4) / torch.where(lab == 0, m["key_neg"][0].cpu().double(), torch.full((A,), float("nan"), dtype=torch.float64))
    ok = lab == 0
    assert abs(float(((e0[ok] - 1) * (e[ok] - 1)).mean())) < 0.03


def test_philox_draws_emulated(emu_lib):
    _run_philox("cpu")


@pytest.mark.gpu
def test_philox_draws_gpu(hip_lib):
    _run_philox("cuda")


def _run_roi_align(dev):
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(9)
    B, C, P = 2, 8, 7
    hw = [(32, 32), (16, 16), (8, 8)]
    scales = [1 / 4, 1 / 8, 1 / 16]
    feats = [torch.randn(B, C, h, w, generator=g) for h, w in hw]
    rois = torch.cat([_rand_boxes(g, 30, 128, 128, 6, 120), torch.tensor([[-20.0, -20, 30, 30], [100, 100, 160, 170], [5, 5, 5.5, 5.2]])])
    R = rois.shape[0]
    bidx = torch.randint(0, B, (R,), generator=g).int()
    box_lists = [Boxes(rois[bidx == b]) for b in range(B)]
    order = torch.cat([torch.where(bidx == b)[0] for b in range(B)])
    pool = U.ROIPooler(P, scales, 0, "ROIAlignV2", canonical_box_size=56, canonical_level=3)
    fr = [f.clone().requires_grad_(True) for f in feats]
    ref = pool(fr, box_lists)                                                      # (R, C, P, P) in `order`
    lv = det.roi_levels(rois.to(dev), 2, 4, 56.0, 3)
    ref_lv = U.assign_boxes_to_levels([Boxes(rois)], 2, 4, 56, 3)
    assert torch.equal(lv.cpu().long(), ref_lv)
    fn = [f.permute(0, 2, 3, 1).contiguous().to(dev) for f in feats]
    out = det.roi_align_fwd(fn, scales, rois.to(dev), bidx.to(dev), lv, P)         # (R, P, P, C)
    got = out.cpu().permute(0, 3, 1, 2)[order]
    assert (got - ref.detach()).abs().max() < 1e-5
    dout = torch.randn(R, P, P, C, generator=g)
    ref.backward(dout.permute(0, 3, 1, 2)[order])
    dfe = [torch.zeros_like(f) for f in fn]
    det.roi_align_bwd(dfe, scales, rois.to(dev), bidx.to(dev), lv, P, dout.to(dev))
    for d, f in zip(dfe, fr):
        fg = f.grad if f.grad is not None else torch.zeros_like(f)
        assert (d.cpu().permute(0, 3, 1, 2) - fg).abs().max() < 2e-5
    # the owner-computes (deterministic) backward overwrites garbage and gives the same gradients
    dfd = [torch.full_like(f, float("nan")) for f in fn]
    det.roi_align_bwd_det(dfd, scales, rois.to(dev), bidx.to(dev), lv, P, dout.to(dev))
    for d, f in zip(dfd, fr):
        fg = f.grad if f.grad is not None else torch.zeros_like(f)
        assert (d.cpu().permute(0, 3, 1, 2) - fg).abs().max() < 2e-5
    # two gradient tensors in one pass (the box head's for every ROI + the cube head's for the first `first` of every `per_image`)
    # == one pass over their sum; also with the first one absent
    per_image, first = 11, 4                                   # 33 ROIs = 3 blocks
    assert R == 3 * per_image
    # forward with the prefix written to a second tensor in the same pass (round 6) == the slice of the full output, bit for bit
    o1, o2 = det.roi_align_fwd2(fn, scales, rois.to(dev), bidx.to(dev), lv, P, per_image, first)
    assert torch.equal(o1.cpu(), out.cpu())
    assert torch.equal(o2.cpu(), out.cpu().view(3, per_image, P, P, C)[:, :first].reshape(-1, P, P, C))
    d2 = torch.randn((R // per_image) * first, P, P, C, generator=g)
    merged = dout.clone()
    merged.view(-1, per_image, P, P, C)[:, :first] += d2.view(-1, first, P, P, C)
    only2 = torch.zeros_like(dout)
    only2.view(-1, per_image, P, P, C)[:, :first] += d2.view(-1, first, P, P, C)
    for first_grad, want_src in ((dout, merged), (None, only2)):
        want = [torch.zeros_like(f) for f in fn]
        det.roi_align_bwd(want, scales, rois.to(dev), bidx.to(dev), lv, P, want_src.to(dev))
        got2 = [torch.zeros_like(f) for f in fn]
        det.roi_align_bwd(got2, scales, rois.to(dev), bidx.to(dev), lv, P, None if first_grad is None else first_grad.to(dev),
                          dout2=d2.to(dev), per_image=per_image, first=first)
        for a, b in zip(got2, want):
            assert (a - b).abs().max() <= 1e-5 * max(1.0, float(b.abs().max()))     # (atomics: summation order differs)
        got3 = [torch.full_like(f, float("nan")) for f in fn]
        det.roi_align_bwd_det(got3, scales, rois.to(dev), bidx.to(dev), lv, P, None if first_grad is None else first_grad.to(dev),
                              dout2=d2.to(dev), per_image=per_image, first=first)
        for a, b in zip(got3, want):
            assert (a - b).abs().max() <= 1e-5 * max(1.0, float(b.abs().max()))


def _run_roi_align_big(dev):
    """a footprint wider than the separable kernel's LDS tables (144 px) takes the per-sample fallback"""
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(10)
    C, P = 4, 7
    feat = torch.randn(1, C, 150, 150, generator=g)
    rois = torch.tensor([[2.0, 3.0, 149.0, 148.0], [10.0, 20.0, 40.0, 30.0]])
    pool = U.ROIPooler(P, [1.0], 0, "ROIAlignV2", canonical_box_size=56, canonical_level=0)
    fr = feat.clone().requires_grad_(True)
    ref = pool([fr], [Boxes(rois)])
    dout = torch.randn(2, P, P, C, generator=g)
    ref.backward(dout.permute(0, 3, 1, 2))
    lv = torch.zeros(2, dtype=torch.int32, device=dev)
    bidx = torch.zeros(2, dtype=torch.int32, device=dev)
    dfe = [torch.zeros(1, 150, 150, C, device=dev)]
    det.roi_align_bwd(dfe, [1.0], rois.to(dev), bidx, lv, P, dout.to(dev))
    assert (dfe[0].cpu().permute(0, 3, 1, 2) - fr.grad).abs().max() < 2e-5
    dfd = [torch.full((1, 150, 150, C), float("nan"), device=dev)]          # ragged 8 x 8 tiles (150 = 18 * 8 + 6), a 147 px footprint
    det.roi_align_bwd_det(dfd, [1.0], rois.to(dev), bidx, lv, P, dout.to(dev))
    assert (dfd[0].cpu().permute(0, 3, 1, 2) - fr.grad).abs().max() < 2e-5


def _run_box_loss(dev):
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(2)
    R, K = 70, 50
    pred = torch.randn(R, 256, generator=g)
    cls = torch.randint(0, K + 1, (R,), generator=g)
    cls[:10] = torch.randint(0, K, (10,), generator=g)
    cls[-5:] = -2
    prop = _rand_boxes(g, R, 300, 300)
    gtb = _rand_boxes(g, 9, 300, 300)
    gt_row = torch.randint(0, 9, (R,), generator=g)
    val = cls >= 0
    scores = pred[val, : K + 1].clone().requires_grad_(True)
    deltas = pred[val, K + 1: K + 1 + 4 * K].clone().requires_grad_(True)
    ref = O.fast_rcnn_losses(scores, deltas, cls[val], prop[val], gtb[gt_row[val]], K)
    args = (pred.to(dev), K, cls.int().to(dev), prop.to(dev), gtb.to(dev), gt_row.int().to(dev))
    sums = det.box_loss_fwd(*args)
    n = sums[2].item()
    assert n == val.sum().item()
    assert abs(sums[0].item() / n - ref["BoxHead/loss_cls"].item()) < 1e-5
    assert abs(sums[1].item() / n - ref["BoxHead/loss_box_reg"].item()) < 1e-5
    pc = scores.detach().argmax(1)
    assert sums[4].item() == (pc == cls[val]).sum().item()
    (1.5 * ref["BoxHead/loss_cls"] + 0.25 * ref["BoxHead/loss_box_reg"]).backward()
    dpred = det.box_loss_bwd(*args, sums, torch.tensor([1.5]).to(dev), torch.tensor([0.25]).to(dev)).cpu()
    assert (dpred[val, : K + 1] - scores.grad).abs().max() < 1e-7
    assert (dpred[val, K + 1: K + 1 + 4 * K] - deltas.grad).abs().max() < 1e-7
    assert dpred[~val].abs().max() == 0 and dpred[:, 5 * K + 1:].abs().max() == 0


def _rand_rot(g, n):
    q = torch.randn(n, 4, generator=g)
    return U.quaternion_to_matrix(q / q.norm(dim=1, keepdim=True))


CUBE_CONFIGS = {
    "base": {},
    # every non-default switch at least once (MODEL.ROI_CUBE_HEAD.*; roi_heads.py:426-768, cube_head.py:175-185)
    "quat_sigmoid": dict(z_type="sigmoid", dims_priors_func="sigmoid", pose_type="quaternion", allocentric_pose=False, chamfer_pose=False),
    "euler_log": dict(z_type="log", dims_priors_enabled=False, pose_type="euler", virtual_depth=False, inverse_z_weight=True,
                      use_confidence=False, joint=False),
    "quat_log_invz": dict(z_type="log", pose_type="quaternion", inverse_z_weight=True),
    "euler_sigmoid_l1": dict(z_type="sigmoid", pose_type="euler", chamfer_pose=False, use_confidence=False),
    # CLUSTER_BINS > 1 (roi_heads.py:432-442) with the plain depth types and Z_TYPE 'clusters' (:501-522)
    "bins_direct": dict(cluster_bins=4),
    "clusters": dict(z_type="clusters", cluster_bins=5, pose_type="quaternion"),
    # DISENTANGLED_LOSS False (roi_heads.py:606-649, 676-680); only evaluates without dimension priors in the reference
    "entangled_direct": dict(disentangled=False, dims_priors_enabled=False),
    "entangled_sigmoid": dict(disentangled=False, dims_priors_enabled=False, z_type="sigmoid", allocentric_pose=False, pose_type="euler",
                              inverse_z_weight=True),
    "entangled_log": dict(disentangled=False, dims_priors_enabled=False, z_type="log", virtual_depth=False, use_confidence=False, joint=False),
    "entangled_clusters": dict(disentangled=False, dims_priors_enabled=False, z_type="clusters", cluster_bins=3, pose_type="quaternion"),
}


def _run_cube(dev, cfg_name="base"):
    from omni3d_amd.kernels import det
    cfg = dict(CUBE_CONFIGS[cfg_name])
    nb = cfg.pop("cluster_bins", 1)
    mode = det.cube_mode(**cfg)
    assert (mode == det.CUBE_MODE_BASE) == (cfg_name in ("base", "bins_direct"))
    conf, joint = cfg.get("use_confidence", True), cfg.get("joint", True)
    g = torch.Generator().manual_seed(4)
    F_, K, B = 37, 50, 3
    W = det.cube_head_width(mode, nb)
    Pn = W - 5 - nb - int(conf)
    ldh = (W * K + 15) // 16 * 16
    head = torch.randn(F_, ldh, generator=g) * 0.5
    if conf:
        head[:, (5 + nb + Pn) * K: W * K] += 1.0       # uncertainties around 1, some below the 0.01 clip
    head[0, (2 + nb) * K: (5 + nb) * K] = 6.0          # dims logits above the clip(max=5)
    if cfg.get("z_type") == "log":
        head[:, 2 * K: 3 * K] += 2.0          # depths around e^2
    if cfg.get("pose_type") == "quaternion":
        head[1, (5 + nb) * K: (5 + nb) * K + 4 * K: 4] = -0.7   # negative real part -> the copysign branch
    boxes = _rand_boxes(g, F_, 512, 512, 20, 200)
    z_scales = z_stats = clusters = None
    if nb > 1:
        z_scales = torch.sort(torch.rand(K, nb, generator=g) * 250 + 20, dim=1).values
        z_scales[3] = z_scales[3, 0]                                    # equal scale priors: argmin keeps the first
        z_stats = torch.stack((torch.rand(K, nb, generator=g) * 30 + 3, torch.rand(K, nb, generator=g) * 6 + 0.5), dim=2)
        z_stats[::7, :, 1] *= 4                                         # mean - 3 std < 0 -> the clip(0)
        clusters = (nb, z_scales.to(dev), z_stats.to(dev) if cfg.get("z_type") == "clusters" else None)
    cls = torch.randint(0, K, (F_,), generator=g)
    img = torch.randint(0, B, (F_,), generator=g)
    Ks = torch.tensor([[500.0, 510.0, 250.0, 260.0], [400.0, 400.0, 256.0, 256.0], [700.0, 690.0, 300.0, 200.0]])
    v2r = torch.tensor([1.0, 0.8, 1.7])
    priors = torch.rand(K, 2, 3, generator=g) * 2 + 0.5
    priors[:, 1] *= 0.3                       # std: some classes with mean - 3 std < 0 (clipped), some above
    G = 11
    gt3d = torch.cat([torch.rand(G, 2, generator=g) * 512, torch.rand(G, 1, generator=g) * 30 + 2, torch.rand(G, 3, generator=g) * 3 + 0.3,
                      torch.zeros(G, 3)], 1)
    gt3d[0, 2] = 1.5                          # below e: the clip of INVERSE_Z_WEIGHT
    gtpose = _rand_rot(g, G)
    gt_row = torch.randint(0, G, (F_,), generator=g)
    loss_w = (1.0, 0.8, 1.2, 0.9, 1.1)
    hr = head[:, : W * K].clone().requires_grad_(True)
    Kmat = torch.zeros(F_, 3, 3)
    Kmat[:, 0, 0], Kmat[:, 1, 1], Kmat[:, 0, 2], Kmat[:, 1, 2], Kmat[:, 2, 2] = Ks[img, 0], Ks[img, 1], Ks[img, 2], Ks[img, 3], 1.0
    ref, stats, ex = O.cube_losses(hr, K, boxes, cls, Kmat, v2r[img], priors[cls, 0], gt3d[gt_row], gtpose[gt_row],
                                   prior_std=priors[cls, 1], loss_w=loss_w, cluster_bins=nb, z_scales=z_scales, z_stats=z_stats, **cfg)
    args = [t.to(dev) for t in (head, boxes, cls.int(), img.int(), Ks, v2r, priors, gt3d, gtpose.reshape(G, 9), gt_row.int())]
    vals, jac, red = det.cube_loss_fwd(args[0], K, *args[1:], loss_w=loss_w, mode=mode, clusters=clusters)
    names = ["Cube/loss_dims", "Cube/loss_xy", "Cube/loss_z", "Cube/loss_pose", "Cube/loss_joint", "Cube/uncert"]
    assert ("Cube/loss_joint" in ref) == joint and ("Cube/uncert" in ref) == conf
    r = red.cpu()
    for k, nm in enumerate(names):
        if nm in ref:
            assert abs(r[k].item() - ref[nm].item()) < 2e-5 * max(1.0, abs(ref[nm].item())), (nm, r[k].item(), ref[nm].item())
        else:
            assert r[k].item() == 0.0, nm
    for k, nm in zip((12, 13, 14, 15, 16, 17), ("Cube/total_3D_loss", "Cube/z_error", "Cube/dims_error", "Cube/xy_error", "Cube/z_close", "Cube/conf")):
        if nm in stats:
            assert abs(r[k].item() - stats[nm]) < 2e-5 * max(1.0, abs(stats[nm])), nm
    w = torch.tensor([1.0, 0.7, 1.3, 0.9, 1.1, 0.5])
    sum(w[k] * ref[nm] for k, nm in enumerate(names) if nm in ref).backward()
    dhead = det.cube_loss_bwd(vals, jac, red, w.to(dev), args[2], args[1], F_, K, ldh, mode, clusters).cpu()
    scale = hr.grad.abs().max().item()
    assert (dhead[:, : W * K] - hr.grad).abs().max() < 2e-5 * max(1.0, scale), (dhead[:, : W * K] - hr.grad).abs().max()
    assert ldh == W * K or dhead[:, W * K:].abs().max() == 0
    # inference decode
    ratio = torch.tensor([1.0, 2.0, 0.5])
    c3, pose, verts = det.cube_decode(args[0], K, args[1], args[2], args[3], args[4], args[5], ratio.to(dev), args[6], mode, clusters)
    X = ex["cube_z"] * (ex["cube_x"] - Kmat[:, 0, 2]) / Kmat[:, 0, 0]
    assert (c3[:, 0].cpu() - X.detach()).abs().max() < 1e-4 * max(1.0, X.detach().abs().max().item())
    assert (c3[:, 3:6].cpu() - ex["cube_dims"].detach()).abs().max() < 1e-5 * max(1.0, ex["cube_dims"].detach().abs().max().item())
    assert (pose.cpu() - ex["cube_pose"].detach()).abs().max() < 1e-5
    if conf:
        assert (c3[:, 8].cpu() - torch.exp(-ex["cube_uncert"].detach())).abs().max() < 1e-6
    else:       # the reference merges scores with `cube_3D[:, -1]`, the scaled v coordinate, when it has no confidence column
        assert torch.equal(c3[:, 8], c3[:, 7]) and (c3[:, 7].cpu() - ex["cube_y"].detach() * ratio[img]).abs().max() < 1e-3
    cc = torch.cat([torch.stack((X, ex["cube_z"] * (ex["cube_y"] - Kmat[:, 1, 2]) / Kmat[:, 1, 1], ex["cube_z"]), 1), ex["cube_dims"]], 1).detach()
    want = O.get_cuboid_verts(cc, ex["cube_pose"].detach())
    assert (verts.cpu() - want).abs().max() < 1e-4 * max(1.0, want.abs().max().item())
    assert (det.cuboid_corners(cc.to(dev), ex["cube_pose"].detach().reshape(-1, 9).to(dev)).cpu() - want).abs().max() < 1e-5 * max(1.0, want.abs().max().item())


def _run_sgd(dev):
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(8)
    n = 1003
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.1, momentum=0.9, weight_decay=1e-2)
    p, buf = p0.clone().to(dev), torch.zeros(n).to(dev)
    for it in range(3):
        gr = g0 * (it + 1)
        ref.grad = gr.clone()
        opt.step()
        det.sgd_step(p, gr.to(dev), buf, 0.1, 0.9, 0.0, 1e-2, False, first_step=(it == 0))
        assert (p.cpu() - ref.detach()).abs().max() < 1e-6
    flag = torch.zeros(1).to(dev)
    det.nonfinite_any(p, flag)
    assert flag.item() == 0
    p[517] = float("nan")
    det.nonfinite_any(p, flag)
    assert flag.item() == 1
    skip = torch.ones(1).to(dev)
    q = p0.clone().to(dev)
    det.sgd_step(q, g0.to(dev), buf, 0.1, skip_flag=skip)
    assert torch.equal(q.cpu(), p0)


def _run_rpn_post_nms_glue(dev):
    """det.rpn_mask_scores / det.rpn_collect == the element-wise torch formulation of find_top_rpn_proposals' last step"""
    from omni3d_amd.kernels import det, select
    g = torch.Generator().manual_seed(4)
    B, N, P = 3, 700, 300
    boxes = (torch.rand(B, N, 4, generator=g) * 100).to(dev)
    scores = torch.randn(B, N, generator=g).to(dev)
    keep = (torch.rand(B, N, generator=g) > 0.7).int().to(dev)
    keep[1] = 0                                                      # an image whose candidates are all suppressed
    keep[2, :5] = 1
    masked = det.rpn_mask_scores(scores, keep)
    ref_masked = torch.where(keep != 0, scores, torch.full_like(scores, float("-inf")))
    assert torch.equal(masked, ref_masked)
    top_v, top_i = select.topk_rows(masked, P)
    prop, count = det.rpn_collect(boxes, top_v, top_i)
    ok = top_v > float("-inf")
    ref = torch.gather(boxes, 1, top_i.clamp(min=0).long()[:, :, None].expand(-1, -1, 4)) * ok[:, :, None]
    assert torch.equal(prop, ref) and torch.equal(count, ok.sum(dim=1).to(torch.int32))
    assert int(count[1]) == 0 and float(prop[1].abs().max()) == 0.0


def test_rpn_post_nms_glue_emulated(emu_lib):
    _run_rpn_post_nms_glue("cpu")


@pytest.mark.gpu
def test_rpn_post_nms_glue_gpu(hip_lib):
    _run_rpn_post_nms_glue("cuda")


def test_rpn_labels_emulated(emu_lib):
    _run_rpn_labels("cpu", 64, 1.0, 0)
    _run_rpn_labels("cpu", 512, 0.5, 1)      # not enough positives -> negatives get sampled, ignore path


def test_rpn_loss_decode_emulated(emu_lib):
    _run_rpn_loss("cpu")


def test_crowded_image_labels_and_sampling_emulated(emu_lib):
    """300 / 700 ground-truth boxes in one image (the LDS staging of csrc/rpn_roi.hip holds 1024; it held 256)"""
    _run_rpn_labels("cpu", 64, 0.5, 3, crowded=300)
    _run_roi_sample("cpu", crowded=700)


@pytest.mark.gpu
def test_crowded_image_labels_and_sampling_gpu(hip_lib):
    _run_rpn_labels("cuda", 64, 0.5, 3, crowded=300)
    _run_roi_sample("cuda", crowded=700)


def test_roi_sample_emulated(emu_lib):
    _run_roi_sample("cpu")


def test_roi_align_emulated(emu_lib):
    _run_roi_align("cpu")
    _run_roi_align_big("cpu")


def test_box_loss_emulated(emu_lib):
    _run_box_loss("cpu")


@pytest.mark.parametrize("cfg_name", sorted(CUBE_CONFIGS))
def test_cube_emulated(emu_lib, cfg_name):
    _run_cube("cpu", cfg_name)


def test_sgd_emulated(emu_lib):
    _run_sgd("cpu")


# one GPU test per kernel family: with `-x` a failure names its kernel and does not hide the other rows
@pytest.mark.gpu
@pytest.mark.parametrize("batch_per_image,pos_frac,seed", [(64, 1.0, 0), (512, 0.5, 1)])
def test_rpn_labels_gpu(hip_lib, batch_per_image, pos_frac, seed):
    _run_rpn_labels("cuda", batch_per_image, pos_frac, seed)


@pytest.mark.gpu
def test_rpn_loss_decode_gpu(hip_lib):
    _run_rpn_loss("cuda")


@pytest.mark.gpu
def test_roi_sample_gpu(hip_lib):
    _run_roi_sample("cuda")


@pytest.mark.gpu
def test_roi_align_gpu(hip_lib):
    _run_roi_align("cuda")


@pytest.mark.gpu
def test_roi_align_big_gpu(hip_lib):
    _run_roi_align_big("cuda")


@pytest.mark.gpu
def test_box_loss_gpu(hip_lib):
    _run_box_loss("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", sorted(CUBE_CONFIGS))
def test_cube_gpu(hip_lib, cfg_name):
    _run_cube("cuda", cfg_name)


@pytest.mark.gpu
def test_sgd_gpu(hip_lib):
    _run_sgd("cuda")


@pytest.mark.parametrize("num_fc", [1, 2, 3])
def test_box_head_num_fc_emulated(emu_lib, num_fc):
    """FastRCNNConvFCHead with MODEL.ROI_BOX_HEAD.NUM_FC 1 / 2 (Base.yaml) / 3 against the restated detectron2 head: names, outputs, gradients"""
    from oracle import make_golden as MG
    from omni3d_amd.cubercnn.modeling.roi_heads.roi_heads import FastRCNNConvFCHead
    from omni3d_amd.d2.layers import ShapeSpec
    cfg = MG.product_cfg(["MODEL.ROI_BOX_HEAD.NUM_FC", num_fc, "MODEL.ROI_BOX_HEAD.FC_DIM", 64])
    torch.manual_seed(num_fc)
    prod = FastRCNNConvFCHead(cfg, ShapeSpec(channels=16, height=7, width=7))
    ref = U.FastRCNNConvFCHead(U.ShapeSpec(channels=16, height=7, width=7), conv_dims=[], fc_dims=[64] * num_fc)
    assert list(prod.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict(prod.state_dict(), strict=True)
    x = torch.randn(9, 16, 7, 7, generator=torch.Generator().manual_seed(1))
    xr = x.clone().requires_grad_(True)
    xp = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yp, yr = prod(xp), ref(xr)
    (yp ** 2).sum().backward()
    (yr ** 2).sum().backward()
    assert (yp - yr).abs().max() < 1e-5 and (xp.grad - xr.grad).abs().max() < 1e-5
    rg = dict(ref.named_parameters())
    for n, p in prod.named_parameters():
        g = p.grad.contiguous(memory_format=torch.contiguous_format).reshape(rg[n].grad.shape)
        assert (g - rg[n].grad).abs().max() < 1e-4 * max(1.0, float(rg[n].grad.abs().max())), n


def _run_roi_align_legacy(dev):
    """POOLER_TYPE "ROIAlign" (round 6): torchvision roi_align with aligned=False -- no half-pixel shift, ROI sides of at least one pixel --
    forward and backward of the product's pooler against the restated torchvision op, level by level like detectron2's ROIPooler"""
    from omni3d_amd.cubercnn.modeling.roi_heads.roi_heads import ROIPooler
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(19)
    B, C, P = 2, 8, 7
    hw = [(32, 32), (16, 16), (8, 8)]
    scales = [1 / 4, 1 / 8, 1 / 16]
    feats = [torch.randn(B, C, h, w, generator=g) for h, w in hw]
    rois = torch.cat([_rand_boxes(g, 30, 128, 128, 6, 120), torch.tensor([[-20.0, -20, 30, 30], [100, 100, 160, 170], [5, 5, 5.5, 5.2], [40, 40, 40, 40]])])
    R = rois.shape[0]
    bidx = torch.randint(0, B, (R,), generator=g).int()
    lv = U.assign_boxes_to_levels([Boxes(rois)], 2, 4, 56, 3)
    fr = [f.clone().requires_grad_(True) for f in feats]
    ref = torch.zeros(R, C, P, P)
    for level, scale in enumerate(scales):
        inds = torch.where(lv == level)[0]
        if inds.numel():
            fmt = torch.cat([bidx[inds, None].float(), rois[inds]], dim=1)
            ref = ref.index_put((inds,), U.roi_align(fr[level], fmt, P, scale, 0, False))
    pool = ROIPooler(P, scales, 0, "ROIAlign", canonical_box_size=56, canonical_level=3)
    assert not pool.aligned and not pool.same_as(pool)
    fd = [f.clone().contiguous(memory_format=torch.channels_last).to(dev).requires_grad_(True) for f in feats]
    got = pool(fd, rois.to(dev), bidx.to(dev))
    assert (got.detach().cpu() - ref.detach()).abs().max() < 1e-5
    # the aligned form differs visibly on the same boxes (the test would not notice a silently ignored flag otherwise)
    al = ROIPooler(P, scales, 0, "ROIAlignV2", canonical_box_size=56, canonical_level=3)(fd, rois.to(dev), bidx.to(dev))
    assert (al.detach().cpu() - ref.detach()).abs().max() > 1e-2
    dout = torch.randn(R, C, P, P, generator=g)
    ref.backward(dout)
    got.backward(dout.to(dev))
    for d, f in zip(fd, fr):
        fg = f.grad if f.grad is not None else torch.zeros_like(f)
        assert (d.grad.cpu() - fg).abs().max() < 5e-5
    with pytest.raises(NotImplementedError):
        ROIPooler(P, scales, 0, "ROIAlignRotated")


def test_roi_align_legacy_pooler_emulated(emu_lib):
    _run_roi_align_legacy("cpu")


@pytest.mark.gpu
def test_roi_align_legacy_pooler_gpu(hip_lib):
    _run_roi_align_legacy("cuda")


def _run_roi_pool(dev):
    """POOLER_TYPE "ROIPool" (round 6): torchvision roi_pool level by level like detectron2's ROIPooler -- forward, argmax routing of the
    gradient (overlapping ROIs add), whole-pixel rounding (x.5 corners), ROIs partly / wholly outside the map (empty bins: 0, no
    gradient), a degenerate ROI (forced to 1 x 1), bins narrower than a pixel (P larger than the ROI: neighbouring bins share a pixel)"""
    from omni3d_amd.cubercnn.modeling.roi_heads.roi_heads import ROIPooler
    g = torch.Generator().manual_seed(23)
    B, C = 2, 8
    hw = [(32, 32), (16, 16), (8, 8)]
    scales = [1 / 4, 1 / 8, 1 / 16]
    feats = [torch.randn(B, C, h, w, generator=g) for h, w in hw]
    rois = torch.cat([_rand_boxes(g, 30, 128, 128, 6, 120),
                      torch.tensor([[-20.0, -20, 30, 30], [100, 100, 160, 170], [5, 5, 5.5, 5.2], [40, 40, 40, 40], [2, 2, 10, 10], [6, 10, 18, 22],
                                    [200, 200, 260, 260], [-50, -50, -10, -10], [0, 0, 127, 127]])])
    R = rois.shape[0]
    bidx = torch.randint(0, B, (R,), generator=g).int()
    lv = U.assign_boxes_to_levels([Boxes(rois)], 2, 4, 56, 3)
    for P in (7, 3):
        fr = [f.clone().requires_grad_(True) for f in feats]
        ref = torch.zeros(R, C, P, P)
        for level, scale in enumerate(scales):
            inds = torch.where(lv == level)[0]
            if inds.numel():
                fmt = torch.cat([bidx[inds, None].float(), rois[inds]], dim=1)
                ref = ref.index_put((inds,), U.roi_pool(fr[level], fmt, P, scale))
        pool = ROIPooler(P, scales, 2, "ROIPool", canonical_box_size=56, canonical_level=3)       # (roi_pool has no sampling ratio: any value passes)
        assert pool.max_pool and not pool.same_as(pool)
        fd = [f.clone().contiguous(memory_format=torch.channels_last).to(dev).requires_grad_(True) for f in feats]
        got = pool(fd, rois.to(dev), bidx.to(dev))
        assert torch.equal(got.detach().cpu(), ref.detach()), P               # a maximum is a copy: exact
        assert bool((ref.detach().reshape(R, -1).abs().sum(1) == 0).any())    # (the wholly-outside ROIs really are empty)
        dout = torch.randn(R, C, P, P, generator=g)
        ref.backward(dout)
        got.backward(dout.to(dev))
        for d, f in zip(fd, fr):
            fg = f.grad if f.grad is not None else torch.zeros_like(f)
            assert (d.grad.cpu() - fg).abs().max() < 1e-5, P                  # (sums of a few gradients per pixel: order differs)


def test_roi_pool_pooler_emulated(emu_lib):
    _run_roi_pool("cpu")


@pytest.mark.gpu
def test_roi_pool_pooler_gpu(hip_lib):
    _run_roi_pool("cuda")
