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


def compute_priors(cfg, datasets, max_cluster_rounds=1000, min_points_for_std=5):
    """math_util.py:292-470 computes per-category (w, h, l) mean / std from the Omni3D annotation index.  Here `datasets` is a
    list of dataset dicts (synthetic / pre-registered); category ids index the table.  -> {'priors_dims_per_cat': K x 2 x 3,
    'priors_bins': None} (CLUSTER_BINS 1, the hot-path configuration)."""
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    if not isinstance(datasets, (list, tuple)):
        raise NotImplementedError("priors from the Omni3D annotation index (datasets.getAnnIds / loadAnns) are dataset plumbing, "
                                  "out of the hot-path scope; pass the list of dataset dicts")
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
