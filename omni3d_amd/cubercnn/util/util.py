"""The small host helpers tools/train_net.py reaches through `cubercnn.util` (:328-329, :363-369, :424)."""
import json
import os

import numpy as np


def file_parts(file_path):
    base_path, tail = os.path.split(file_path)
    name, ext = os.path.splitext(tail)
    return base_path, name, ext


def save_json(path, data):
    with open(path, "w") as fp:
        json.dump(data, fp)


def load_json(path):
    with open(path, "r") as fp:
        return json.load(fp)


class CubeRCNNHandler:
    """model-zoo path handler (util/model_zoo.py): `cubercnn://...` -> https://dl.fbaipublicfiles.com/cubercnn/...; there is
    no network on the MI355X image, so only an already-downloaded local copy resolves"""
    PREFIX = "cubercnn://"
    CUBERCNN_PREFIX = "https://dl.fbaipublicfiles.com/cubercnn/"

    def _get_supported_prefixes(self):
        return [self.PREFIX]

    def _get_local_path(self, path):
        name = path[len(self.PREFIX):]
        local = os.path.join(os.environ.get("OMNI3D_MODEL_ZOO", os.path.expanduser("~/.torch/iopath_cache/cubercnn")), name)
        if not os.path.exists(local):
            raise FileNotFoundError(f"{self.CUBERCNN_PREFIX + name} is not cached at {local} (no network on this image)")
        return local


def approx_eval_resolution(h, w, scale_min=0, scale_max=1e10):
    """math_util.py:262-289: the resolution an h x w image is run at under ResizeShortestEdge(scale_min, scale_max)
    -> (h, w, factor original -> network)"""
    orig_h = h
    sf = scale_min / min(h, w)
    h, w = h * sf, w * sf
    sf = min(scale_max / max(h, w), 1.0)
    h, w = h * sf, w * sf
    return h, w, h / orig_h


def _mean_std(values):
    """pandas Series.mean() / .std() (ddof 1; NaN for a single sample) without pandas"""
    a = np.asarray(values, dtype=np.float64)
    return float(a.mean()), (float(a.std(ddof=1)) if len(a) > 1 else float("nan"))


def compute_priors(cfg, datasets, max_cluster_rounds=1000, min_points_for_std=5):
    """math_util.py:292-493: per-category statistics of the training annotations.  `datasets`: the Omni3D annotation index
    (`cubercnn.data.Omni3D`: getAnnIds / loadAnns / imgs) or, as an extension, a list of dataset dicts.  Non-ignored annotations
    of the model's categories (`MetadataCatalog.get('omni3d_model').thing_classes`) contribute; -> {'priors_dims_per_cat'
    (K x [mean (w,h,l), std (w,h,l)]), 'priors_z3d_per_cat', 'priors_y3d_per_cat', 'priors_z3d', 'priors_y3d', 'priors_bins'};
    categories without samples get the reference's placeholders ([[1,1,1],[1,1,1]], [50,50], [1,10]).  Depth is expressed in
    virtual units when MODEL.ROI_CUBE_HEAD.VIRTUAL_DEPTH (as the reference does for its z statistics).  With CLUSTER_BINS > 1
    every category also gets `priors_bins` = (name, [2D scale of each cluster], [[z mean, z std] of each cluster]) from a 1-D
    k-means over the box diagonal at test resolution (:401-485)."""
    if isinstance(datasets, (list, tuple)):
        return _priors_from_dicts(cfg, datasets, min_points_for_std)
    from ...d2.data import MetadataCatalog
    from ...d2.structures import BoxMode
    names = list(MetadataCatalog.get("omni3d_model").thing_classes)
    c = cfg.MODEL.ROI_CUBE_HEAD
    rows = {n: [] for n in names}          # name -> [(y3d, z3d, w3d, h3d, l3d, 2D scale)]
    for ann in datasets.loadAnns(datasets.getAnnIds()):
        name = ann["category_name"].lower()
        im = datasets.imgs[ann["image_id"]]
        fy, im_h, im_w = im["K"][1][1], im["height"], im["width"]
        if cfg.DATASETS.MODAL_2D_BOXES and "bbox2D_tight" in ann and ann["bbox2D_tight"][0] != -1:
            box = ann["bbox2D_tight"]
        elif cfg.DATASETS.TRUNC_2D_BOXES and "bbox2D_trunc" in ann and not all(v == -1 for v in ann["bbox2D_trunc"]):
            box = ann["bbox2D_trunc"]
        elif "bbox2D_proj" in ann:
            box = ann["bbox2D_proj"]
        else:
            continue                            # no usable 2D box: the reference skips the annotation here as well
        _, y3d, z3d = ann["center_cam"]
        w3d, h3d, l3d = ann["dimensions"]
        test_h, _, sf = approx_eval_resolution(im_h, im_w, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
        scale = float(np.sqrt(((box[3] - box[1]) * sf) ** 2 + ((box[2] - box[0]) * sf) ** 2))      # diagonal at test resolution (:344-353)
        if c.VIRTUAL_DEPTH:
            z3d = z3d * (1 / ((test_h * fy) / (c.VIRTUAL_FOCAL * im_h)))  # real -> virtual (math_util.py:349-351, 581-592)
        if not ann["ignore"] and name in rows:
            rows[name].append((y3d, z3d, w3d, h3d, l3d, scale))
    everything = [r for n in names for r in rows[n]]
    if everything:
        arr = np.asarray(everything, dtype=np.float64)
        priors_y3d, priors_z3d = list(_mean_std(arr[:, 0])), list(_mean_std(arr[:, 1]))
    else:
        priors_y3d, priors_z3d = [float("nan")] * 2, [float("nan")] * 2
    dims, z_cat, y_cat, bins = [], [], [], []
    n_bins = c.CLUSTER_BINS
    for n in names:
        if n_bins > 1:
            arr = np.asarray(rows[n], dtype=np.float64).reshape(-1, 6)
            bins.append((n,) + _scale_clusters(arr[:, 5], arr[:, 1], n_bins, max_cluster_rounds, min_points_for_std,
                                               cfg.MODEL.ANCHOR_GENERATOR.SIZES))
        if rows[n]:
            arr = np.asarray(rows[n], dtype=np.float64)
            ms = [_mean_std(arr[:, k]) for k in range(5)]
            dims.append([[ms[2][0], ms[3][0], ms[4][0]], [ms[2][1], ms[3][1], ms[4][1]]])
            z_cat.append(list(ms[1]))
            y_cat.append(list(ms[0]))
        else:
            dims.append([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]])
            z_cat.append([50, 50])
            y_cat.append([1, 10])
    return {"priors_dims_per_cat": dims, "priors_z3d_per_cat": z_cat, "priors_y3d_per_cat": y_cat, "priors_bins": bins,
            "priors_y3d": priors_y3d, "priors_z3d": priors_z3d}


def _scale_clusters(scales, z3d, n_bins, max_rounds, min_points, anchor_sizes):
    """math_util.py:401-485 for one category: 1-D k-means of the 2D box scale (float32, like the reference's FloatTensor), clusters
    initialised geometrically between the smallest and the largest scale, a round accepted while the mean |scale - centre|
    (rounded to 5 decimals) improves; under-populated clusters borrow their `min_points` nearest samples.
    -> ([cluster scale], [[z mean, z std]])"""
    n = len(scales)
    if n < min_points:                          # placeholders: anchors' geometric range, depths from 100 down (:423-437)
        print("Warning category has only {} valid samples...".format(n))
        lo, hi = anchor_sizes[0][0], anchor_sizes[-1][-1]
        base = (hi / lo) ** (1 / (n_bins - 1))
        zs = [[b, 15] for b in np.arange(100, 1, -(100 - 1) / n_bins)]
        assert len(zs) == n_bins, "Broken default bin scaling."
        return [float(lo * (base ** i)) for i in range(n_bins)], zs
    s = np.asarray(scales, dtype=np.float32)
    lo, hi = s.min(), s.max()
    base = (hi / lo) ** np.float32(1 / (n_bins - 1))
    centres = np.asarray([lo * (base ** i) for i in range(n_bins)], dtype=np.float32)

    def members(assign, quality, b):
        m = assign == b
        if m.sum() < min_points:                # not enough samples: add the min_points nearest ones
            m = m.copy()
            m[np.argsort(-quality[:, b], kind="stable")[:min_points]] = True
        return m

    best, assign = -np.inf, None
    for _ in range(max_rounds):
        quality = -np.abs(centres[None, :] - s[:, None])
        this = quality.argmax(1)
        score = float(quality.max(1).mean(dtype=np.float32))
        if np.round(score, 5) > best:
            best, assign = score, this
            centres = np.asarray([s[members(assign, quality, b)].mean(dtype=np.float32) for b in range(n_bins)], dtype=np.float32)
        else:
            break
    z = np.asarray(z3d, dtype=np.float64)
    return [float(v) for v in centres], [list(_mean_std(z[members(assign, quality, b)])) for b in range(n_bins)]


def _priors_from_dicts(cfg, datasets, min_points_for_std=5):
    """extension: the same per-category dimension statistics from a list of dataset dicts (synthetic / pre-registered data)"""
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    dims = [[] for _ in range(K)]
    for d in datasets:
        for a in d.get("annotations", []):
            c = int(a["category_id"])
            if 0 <= c < K and not a.get("ignore", False):
                dims[c].append(a["dimensions"])
    out = []
    for c in range(K):
        if len(dims[c]) > 0:
            arr = np.asarray(dims[c], dtype=np.float64)
            std = arr.std(axis=0, ddof=1) if len(arr) >= min_points_for_std else np.ones(3)
            out.append([arr.mean(axis=0).tolist(), std.tolist()])
        else:
            out.append([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]])           # math_util.py:396-399 dummy data
    return {"priors_dims_per_cat": out, "priors_bins": None}
