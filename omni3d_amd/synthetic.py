"""Synthetic Omni3D-shaped training batches (BASELINE configs 1-4).

Produces the batched-input schema of the reference's DatasetMapper3D
(/root/reference/cubercnn/data/dataset_mapper.py:17-58,133-155): per image a dict with `image`
uint8 (3,H,W) BGR, `height`, `width`, `K` 3x3, `image_id`, `dataset_id`, and `instances` carrying
gt_classes int64 (-1 = ignore), gt_boxes (Boxes XYXY), gt_boxes3D (M,9) = [u, v, z, w, h, l, X, Y, Z]
and gt_poses (M,3,3).  There is no network and no dataset here; benchmarks and parity tests use this
generator (seeded, numpy only) on every side of a comparison.
"""
import numpy as np
import torch

from .d2.structures import Boxes, Instances


def make_priors(num_classes=50, seed=0, bins=0):
    """`priors['priors_dims_per_cat']` (K,2,3): per-class mean / std of (w,h,l) in metres
    (roi_heads.py:117-118).  Fixed table: mean U[.3,4], std = 0.2 * mean.  bins > 1 adds `priors_bins` in the form
    math_util.compute_priors emits for CLUSTER_BINS (:436,485): per class (name, [2D scale of each cluster], [[z mean, z std]])."""
    rs = np.random.RandomState(1000 + seed)
    mean = rs.uniform(0.3, 4.0, size=(num_classes, 3))
    std = 0.2 * mean
    out = {"priors_dims_per_cat": np.stack([mean, std], axis=1).astype(np.float32).tolist(), "priors_bins": None}
    if bins and bins > 1:
        out["priors_bins"] = []
        for c in range(num_classes):
            scales = np.sort(rs.uniform(10.0, 200.0, size=bins))
            zmean = np.sort(rs.uniform(3.0, 40.0, size=bins))[::-1]            # large on screen = close
            zstd = rs.uniform(0.5, 8.0, size=bins)
            out["priors_bins"].append((f"class{c}", scales.astype(np.float32).tolist(),
                                       np.stack([zmean, zstd], axis=1).astype(np.float32).tolist()))
    return out


def _cuboid_corners(xyz, whl, R):
    """8 corners in the order of math_util.get_cuboid_verts_faces (math_util.py:151-181)."""
    w, h, l = whl
    x = np.array([-l, l, l, -l, -l, l, l, -l]) / 2
    y = np.array([-h, -h, h, h, -h, -h, h, h]) / 2
    z = np.array([-w, -w, -w, -w, w, w, w, w]) / 2
    v = R @ np.stack([x, y, z])
    return (v + np.asarray(xyz)[:, None]).T


def _euler_yxz(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


def make_batch(num_images=2, height=512, width=512, num_gt=8, num_classes=50, seed=0, priors=None, focal=None,
               num_ignore=1):
    priors = priors or make_priors(num_classes)
    prior_mean = np.asarray(priors["priors_dims_per_cat"])[:, 0, :]
    rs = np.random.RandomState(seed)
    f = float(focal if focal is not None else max(height, width))
    K = [[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]]
    batch = []
    for i in range(num_images):
        image = torch.from_numpy(rs.randint(0, 256, size=(3, height, width)).astype(np.uint8))
        classes, boxes2d, boxes3d, poses = [], [], [], []
        for j in range(num_gt):
            c = int(rs.randint(0, num_classes))
            whl = prior_mean[c] * rs.uniform(0.8, 1.2, size=3)
            z = rs.uniform(2.0, 40.0)
            u = rs.uniform(0.15 * width, 0.85 * width)
            v = rs.uniform(0.15 * height, 0.85 * height)
            X = z * (u - K[0][2]) / f
            Y = z * (v - K[1][2]) / f
            R = _euler_yxz(rs.uniform(-np.pi, np.pi), rs.normal(scale=0.05), rs.normal(scale=0.05))
            corners = _cuboid_corners((X, Y, z), whl, R)
            zc = np.maximum(corners[:, 2], 0.1)
            px = f * corners[:, 0] / zc + K[0][2]
            py = f * corners[:, 1] / zc + K[1][2]
            x1, y1 = max(px.min(), 0.0), max(py.min(), 0.0)
            x2, y2 = min(px.max(), float(width)), min(py.max(), float(height))
            if x2 - x1 < 2.0:
                x1, x2 = max(u - 1.0, 0.0), min(u + 1.0, float(width))
            if y2 - y1 < 2.0:
                y1, y2 = max(v - 1.0, 0.0), min(v + 1.0, float(height))
            classes.append(-1 if j >= num_gt - num_ignore else c)
            boxes2d.append([x1, y1, x2, y2])
            boxes3d.append([u, v, z, whl[0], whl[1], whl[2], X, Y, z])
            poses.append(R)
        inst = Instances((height, width))
        inst.gt_classes = torch.tensor(classes, dtype=torch.int64)
        inst.gt_boxes = Boxes(torch.tensor(np.asarray(boxes2d), dtype=torch.float32))
        inst.gt_boxes3D = torch.tensor(np.asarray(boxes3d), dtype=torch.float32)
        inst.gt_poses = torch.tensor(np.asarray(poses), dtype=torch.float32)
        batch.append({"image": image, "height": height, "width": width, "K": K, "image_id": seed * 1000 + i,
                      "dataset_id": 0, "instances": inst})
    return batch


def make_dataset_dicts(num_images=8, height=96, width=128, num_gt=4, num_classes=50, seed=0, priors=None):
    """The same synthetic scenes as `make_batch`, in the DATASET-dict schema the reference's loader feeds to DatasetMapper3D
    (cubercnn/data/datasets.py:230-330): file-level fields + `annotations` with bbox (XYXY_ABS), category_id, center_cam,
    dimensions, pose / R_cam, bbox3D_cam (8 corners), ignore, iscrowd.  Pixels travel in memory (`image_array`, HWC BGR)."""
    from .d2.structures import BoxMode
    out = []
    for i, b in enumerate(make_batch(num_images, height, width, num_gt, num_classes, seed, priors)):
        inst = b["instances"]
        annos = []
        for j in range(len(inst)):
            c = int(inst.gt_classes[j])
            g = inst.gt_boxes3D[j].tolist()
            R = inst.gt_poses[j].numpy().astype(np.float64)
            annos.append({"bbox": inst.gt_boxes.tensor[j].tolist(), "bbox_mode": BoxMode.XYXY_ABS, "category_id": max(c, 0),
                          "center_cam": g[6:9], "dimensions": g[3:6], "pose": R.tolist(), "R_cam": R.tolist(),
                          "bbox3D_cam": _cuboid_corners(g[6:9], g[3:6], R).tolist(), "ignore": c < 0, "iscrowd": 0,
                          "category_name": f"class{max(c, 0)}"})
        out.append({"image_array": np.ascontiguousarray(b["image"].numpy().transpose(1, 2, 0)), "height": height, "width": width,
                    "K": b["K"], "image_id": b["image_id"], "dataset_id": 0, "annotations": annos, "file_name": f"synthetic://{seed}/{i}"})
    return out


def register_synthetic_dataset(name, **kw):
    from .d2.data import DatasetCatalog
    dicts = make_dataset_dicts(**kw)
    if name in DatasetCatalog:
        DatasetCatalog.remove(name)
    DatasetCatalog.register(name, lambda: dicts)
    return dicts


def write_omni3d_dataset(root, name, category_names, category_ids=None, num_images=4, height=96, width=128, num_gt=4, seed=0, dataset_id=0,
                         source="synthetic", image_id_base=None):
    """Writes a synthetic split in the Omni3D ON-DISK format (reference DATA.md:133-198) under `root`:
    `root/datasets/Omni3D/<name>.json` + `root/datasets/<name>/images/*.png`, the layout `cubercnn.data.simple_register` and
    tools/train_net.py:380 expect relative to the working directory.  Scenes are the ones of `make_batch` (boxes projected from
    the 3D cuboids, so `bbox2D_proj` / `bbox2D_trunc` / `bbox3D_cam` / `center_cam` / `dimensions` / `R_cam` are consistent),
    categories cycle through `category_names`.  -> path of the JSON file."""
    import json
    import os
    from PIL import Image
    category_ids = list(category_ids) if category_ids is not None else list(range(len(category_names)))
    img_dir = os.path.join(root, "datasets", name, "images")
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(os.path.join(root, "datasets", "Omni3D"), exist_ok=True)
    base = image_id_base if image_id_base is not None else (dataset_id + 1) * 100000
    images, annos = [], []
    for i, b in enumerate(make_batch(num_images, height, width, num_gt, len(category_names), seed, None)):
        rel = os.path.join(name, "images", f"{i:06d}.png")
        Image.fromarray(np.ascontiguousarray(b["image"].numpy().transpose(1, 2, 0)[:, :, ::-1])).save(os.path.join(root, "datasets", rel))   # BGR -> RGB file
        img_id = base + i
        images.append({"id": img_id, "dataset_id": dataset_id, "width": width, "height": height, "file_path": rel, "K": b["K"],
                       "src_90_rotate": 0, "src_flagged": False})
        inst = b["instances"]
        for j in range(len(inst)):
            k = (i * num_gt + j) % len(category_names)
            g = inst.gt_boxes3D[j].tolist()
            R = inst.gt_poses[j].numpy().astype(np.float64)
            box = [float(v) for v in inst.gt_boxes.tensor[j].tolist()]
            annos.append({"id": len(annos) + 1 + base * 100, "image_id": img_id, "dataset_id": dataset_id, "category_id": category_ids[k],
                          "category_name": category_names[k], "valid3D": True, "bbox2D_tight": [-1, -1, -1, -1], "bbox2D_proj": box,
                          "bbox2D_trunc": box, "bbox3D_cam": _cuboid_corners(g[6:9], g[3:6], R).tolist(), "center_cam": [float(v) for v in g[6:9]],
                          "dimensions": [float(v) for v in g[3:6]], "R_cam": R.tolist(), "behind_camera": False, "visibility": 1.0,
                          "truncation": 0.0, "segmentation_pts": -1, "lidar_pts": -1, "depth_error": -1})
    data = {"info": {"id": dataset_id, "source": source, "name": name, "split": name.rpartition("_")[2], "version": "0.1", "url": ""},
            "images": images, "categories": [{"id": cid, "name": n, "supercategory": "object"} for cid, n in zip(category_ids, category_names)],
            "annotations": annos}
    path = os.path.join(root, "datasets", "Omni3D", name + ".json")
    with open(path, "w") as f:
        json.dump(data, f)
    return path


def write_omni3d_stats(root, category_names, category_ids=None):
    """`datasets/Omni3D/stats.json` with the fields `register_and_store_model_metadata` reads (`category_names`, `categories`)"""
    import json
    import os
    category_ids = list(category_ids) if category_ids is not None else list(range(len(category_names)))
    path = os.path.join(root, "datasets", "Omni3D", "stats.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"n_datasets": 1, "n_ims": 0, "n_anns": 0, "category_names": list(category_names),
                   "categories": [{"id": cid, "name": n} for cid, n in zip(category_ids, category_names)]}, f)
    return path


# ---- batches shaped like the reference's real input regime ----------------------------------------------------------------------
# configs/Base.yaml:10-13: INPUT.MIN_SIZE_TRAIN = 256, 272, ..., 640 (25 values), MAX_SIZE_TRAIN 4096; cubercnn/data/dataset_mapper.py
# :17-58 resizes every image to a short edge drawn from that list.  Aspect ratios of the Omni3D sources: KITTI 1242 x 375, nuScenes
# 1600 x 900, SUN RGB-D 730 x 530 / Hypersim 1024 x 768, Objectron 1440 x 1920 (portrait).
MIN_SIZE_TRAIN = tuple(range(256, 641, 16))
ASPECTS = ((375, 1242), (900, 1600), (530, 730), (768, 1024), (1920, 1440))


def make_multiscale_batch(num_images, seed, priors=None, num_gt=8, max_long_edge=1344):
    """`num_images` synthetic images, each with its own short edge (MIN_SIZE_TRAIN) and source aspect ratio, like one iteration of the
    reference's loader; max_long_edge bounds the synthetic KITTI-shaped images (2120 px at short edge 640 would only make the
    benchmark's buckets bigger, not different)"""
    rs = np.random.RandomState(seed)
    batch = []
    for i in range(num_images):
        short = int(MIN_SIZE_TRAIN[rs.randint(len(MIN_SIZE_TRAIN))])
        ah, aw = ASPECTS[rs.randint(len(ASPECTS))]
        if ah <= aw:
            h, w = short, int(round(short * aw / ah))
        else:
            h, w = int(round(short * ah / aw)), short
        if max(h, w) > max_long_edge:
            sc = max_long_edge / max(h, w)
            h, w = int(round(h * sc)), int(round(w * sc))
        batch += make_batch(1, h, w, num_gt=num_gt, seed=seed * 131 + i, priors=priors)
    return batch
