"""Omni3D annotation files -> registered datasets (`cubercnn.data.datasets` of the reference,
/root/reference/cubercnn/data/datasets.py:22-448, as imported by tools/train_net.py:33-40 and used at :338-387).

The reference builds on pycocotools' `COCO` class; here the index is a small self-contained class (`Omni3D`) that offers the
COCO calls this code base makes (getAnnIds / getCatIds / getImgIds / loadAnns / loadCats / loadImgs / loadRes, `.dataset`,
`.anns`, `.imgs`, `.cats`, `.imgToAnns`, `.catToImgs`).  JSON layout: DATA.md:133-198 of the reference.  CPU-side plumbing:
nothing here touches the GPU."""
import copy
import json
import logging
import os
from collections import defaultdict

import numpy as np

from ...d2.data import DatasetCatalog, MetadataCatalog
from ...d2.structures import BoxMode
from .. import util

VERSION = "0.1"
logger = logging.getLogger(__name__)
_DEFAULT_ROOT = os.path.join("datasets", "Omni3D")


def get_version():
    return VERSION


# ---- global statistics file (datasets/Omni3D/stats.json, datasets.py:25-49) ---------------------------------------------
def get_global_dataset_stats(path_to_stats=None, reset=False):
    path = path_to_stats or os.path.join(_DEFAULT_ROOT, "stats.json")
    if os.path.exists(path) and not reset:
        return util.load_json(path)
    return {"n_datasets": 0, "n_ims": 0, "n_anns": 0, "categories": []}


def save_global_dataset_stats(stats, path_to_stats=None):
    util.save_json(path_to_stats or os.path.join(_DEFAULT_ROOT, "stats.json"), stats)


def get_filter_settings_from_cfg(cfg=None):
    """datasets.py:52-79: the annotation filter built from cfg.DATASETS (max_height_thres is fixed at 1.5 there too)"""
    if cfg is None:
        return {"category_names": [], "ignore_names": [], "truncation_thres": 0.99, "visibility_thres": 0.01, "min_height_thres": 0.00,
                "max_height_thres": 1.50, "modal_2D_boxes": False, "trunc_2D_boxes": False, "max_depth": 1e8}
    d = cfg.DATASETS
    return {"category_names": d.CATEGORY_NAMES, "ignore_names": d.IGNORE_NAMES, "truncation_thres": d.TRUNCATION_THRES,
            "visibility_thres": d.VISIBILITY_THRES, "min_height_thres": d.MIN_HEIGHT_THRES, "modal_2D_boxes": d.MODAL_2D_BOXES,
            "trunc_2D_boxes": d.TRUNC_2D_BOXES, "max_depth": d.MAX_DEPTH, "max_height_thres": 1.50}


# ---- which 2D box an annotation contributes, which annotations are "ignore" (datasets.py:82-122) --------------------------
def _has_box(anno, key):
    return key in anno and anno[key][0] != -1


def _has_any(anno, key):
    return key in anno and not all(v == -1 for v in anno[key])


def _xywh(box_xyxy):
    return BoxMode.convert(box_xyxy, BoxMode.XYXY_ABS, BoxMode.XYWH_ABS)


def is_ignore(anno, filter_settings, image_height):
    """True when the annotation may not be used as a training / evaluation target: behind the camera, invalid 3D, degenerate
    dimensions, too far, no lidar / segmentation support, depth error, 2D box too small or too large, truncated, invisible,
    or of an ignore category."""
    if anno["behind_camera"] or not bool(anno["valid3D"]):
        return True
    fs = filter_settings
    bad = any(d <= 0 for d in anno["dimensions"][:3]) or anno["center_cam"][2] > fs["max_depth"]
    bad = bad or anno["lidar_pts"] == 0 or anno["segmentation_pts"] == 0 or anno["depth_error"] > 0.5
    if fs["modal_2D_boxes"] and _has_box(anno, "bbox2D_tight"):
        box = _xywh(anno["bbox2D_tight"])
    elif fs["trunc_2D_boxes"] and _has_any(anno, "bbox2D_trunc"):
        box = _xywh(anno["bbox2D_trunc"])
    elif "bbox2D_proj" in anno:
        box = _xywh(anno["bbox2D_proj"])
    else:
        box = anno["bbox"]
    bad = bad or box[3] <= fs["min_height_thres"] * image_height or box[3] >= fs["max_height_thres"] * image_height
    bad = bad or (anno["truncation"] >= 0 and anno["truncation"] >= fs["truncation_thres"])
    bad = bad or (anno["visibility"] >= 0 and anno["visibility"] <= fs["visibility_thres"])
    if "ignore_names" in fs:
        bad = bad or anno["category_name"] in fs["ignore_names"]
    return bool(bad)


# ---- the annotation index ----------------------------------------------------------------------------------------------------
class Omni3D:
    """COCO-style index over one or several Omni3D annotation files (datasets.py:140-291).

    Several files are concatenated (`dataset['info']` becomes a list, every info gains `known_category_ids`); categories are
    the union ordered by id, restricted to `filter_settings['category_names']` when given (otherwise that list is FILLED IN
    with every category found -- callers rely on this side effect, tools/train_net.py:381-387).  With filter settings every
    annotation gets `area`, `iscrowd`, `ignore` / `ignore2D` / `ignore3D`, `bbox` (XYWH), `bbox3D`, `depth`, annotations without
    any usable 2D box are dropped and so are those whose category is neither trainable nor an ignore name."""

    def __init__(self, annotation_files=None, filter_settings=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_files is None:
            return
        if isinstance(annotation_files, str):
            annotation_files = [annotation_files]
        cat_by_id = {}
        for path in annotation_files:
            with open(path, "r") as f:
                ds = json.load(f)
            assert isinstance(ds, dict), "annotation file format {} not supported".format(type(ds))
            if isinstance(ds["info"], list):
                ds["info"] = ds["info"][0]
            ds["info"]["known_category_ids"] = [c["id"] for c in ds["categories"]]
            if not self.dataset:
                self.dataset = ds
            else:
                if isinstance(self.dataset["info"], dict):
                    self.dataset["info"] = [self.dataset["info"]]
                self.dataset["info"].append(ds["info"])
                self.dataset["annotations"] += ds["annotations"]
                self.dataset["images"] += ds["images"]
            for c in ds["categories"]:
                cat_by_id.setdefault(c["id"], c)
        ordered = [cat_by_id[i] for i in sorted(cat_by_id)]
        if filter_settings is None:
            self.dataset["categories"] = ordered
        else:
            self._apply_filter(ordered, filter_settings)
        self.createIndex()

    def _apply_filter(self, ordered_cats, fs):
        trainable = set(fs["ignore_names"]) | set(fs["category_names"])
        if len(fs["category_names"]) > 0:
            self.dataset["categories"] = [c for c in ordered_cats if c["name"] in fs["category_names"]]
        else:       # no names given: use every category found, and tell the caller (side effect of the reference)
            self.dataset["categories"] = ordered_cats
            fs["category_names"] = [c["name"] for c in ordered_cats]
            trainable |= set(fs["category_names"])
        height_of = {im["id"]: im["height"] for im in self.dataset["images"]}
        kept = []
        for anno in self.dataset["annotations"]:
            ignore = is_ignore(anno, fs, height_of[anno["image_id"]])
            if fs["trunc_2D_boxes"] and _has_any(anno, "bbox2D_trunc"):
                box = _xywh(anno["bbox2D_trunc"])
            elif anno["bbox2D_proj"][0] != -1:
                box = _xywh(anno["bbox2D_proj"])
            elif anno["bbox2D_tight"][0] != -1:
                box = _xywh(anno["bbox2D_tight"])
            else:
                continue
            anno["area"] = box[2] * box[3]
            anno["iscrowd"] = False
            anno["ignore"] = anno["ignore2D"] = anno["ignore3D"] = ignore
            anno["bbox"] = _xywh(anno["bbox2D_tight"]) if (fs["modal_2D_boxes"] and anno["bbox2D_tight"][0] != -1) else box
            anno["bbox3D"] = anno["bbox3D_cam"]
            anno["depth"] = anno["center_cam"][2]
            if anno["category_name"] in trainable:
                kept.append(anno)
        self.dataset["annotations"] = kept

    # ---- pycocotools.coco.COCO surface [upstream, published API] ----
    def createIndex(self):
        self.anns, self.cats, self.imgs = {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        for a in self.dataset.get("annotations", []):
            self.imgToAnns[a["image_id"]].append(a)
            self.anns[a["id"]] = a
            self.catToImgs[a["category_id"]].append(a["image_id"])
        for im in self.dataset.get("images", []):
            self.imgs[im["id"]] = im
        for c in self.dataset.get("categories", []):
            self.cats[c["id"]] = c

    @staticmethod
    def _as_list(v):
        return list(v) if isinstance(v, (list, tuple, set, np.ndarray)) else [v]

    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
        imgIds, catIds = self._as_list(imgIds), self._as_list(catIds)
        if len(imgIds) > 0:
            anns = [a for i in imgIds if i in self.imgToAnns for a in self.imgToAnns[i]]
        else:
            anns = self.dataset.get("annotations", [])
        if len(catIds) > 0:
            cs = set(catIds)
            anns = [a for a in anns if a["category_id"] in cs]
        if len(areaRng) > 0:
            anns = [a for a in anns if areaRng[0] < a["area"] < areaRng[1]]
        if iscrowd is not None:
            anns = [a for a in anns if a["iscrowd"] == iscrowd]
        return [a["id"] for a in anns]

    def getCatIds(self, catNms=[], supNms=[], catIds=[]):
        catNms, supNms, catIds = self._as_list(catNms), self._as_list(supNms), self._as_list(catIds)
        cats = self.dataset.get("categories", [])
        if len(catNms) > 0:
            cats = [c for c in cats if c["name"] in catNms]
        if len(supNms) > 0:
            cats = [c for c in cats if c.get("supercategory") in supNms]
        if len(catIds) > 0:
            cats = [c for c in cats if c["id"] in catIds]
        return [c["id"] for c in cats]

    def getImgIds(self, imgIds=[], catIds=[]):
        imgIds, catIds = self._as_list(imgIds), self._as_list(catIds)
        ids = set(imgIds) if len(imgIds) > 0 else set(self.imgs.keys())
        for i, c in enumerate(catIds):
            ids = set(self.catToImgs[c]) if (i == 0 and len(imgIds) == 0) else ids & set(self.catToImgs[c])
        return list(ids)

    def loadAnns(self, ids=[]):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple, np.ndarray)) else [self.anns[ids]]

    def loadCats(self, ids=[]):
        return [self.cats[i] for i in ids] if isinstance(ids, (list, tuple, np.ndarray)) else [self.cats[ids]]

    def loadImgs(self, ids=[]):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple, np.ndarray)) else [self.imgs[ids]]

    def loadRes(self, results):
        """detections (list of dicts with image_id / category_id / bbox XYWH / score ...) as an index over the same images and
        categories; every detection gets `id`, `area` (from its 2D box) and `iscrowd`"""
        res = Omni3D()
        res.dataset = {"images": list(self.dataset.get("images", [])), "categories": copy.deepcopy(self.dataset.get("categories", []))}
        anns = results if isinstance(results, list) else json.load(open(results))
        assert isinstance(anns, list), "results in not an array of objects"
        known = set(self.getImgIds())
        assert {a["image_id"] for a in anns} <= known, "Results do not correspond to current coco set"
        for k, a in enumerate(anns):
            bb = a["bbox"]
            a["area"] = bb[2] * bb[3]
            a["id"] = k + 1
            a["iscrowd"] = 0
        res.dataset["annotations"] = anns
        res.createIndex()
        return res

    def info(self):
        infos = self.dataset["info"]
        for i, inf in enumerate([infos] if isinstance(infos, dict) else infos):
            print("Dataset {}".format(i + 1))
            for key, value in inf.items():
                print("{}: {}".format(key, value))


# ---- detectron2-style dataset dicts (datasets.py:330-448) -------------------------------------------------------------------
_ANN_KEYS = ("bbox", "bbox3D_cam", "bbox2D_proj", "bbox2D_trunc", "bbox2D_tight", "center_cam", "dimensions", "pose", "R_cam", "category_id")


def load_omni3d_json(json_file, image_root, dataset_name, filter_settings, filter_empty=False):
    """One annotation file -> list of per-image dicts (`file_name`, `height`, `width`, `K`, `image_id`, `dataset_id`,
    `annotations`), category ids mapped through the MODEL's table (`MetadataCatalog.get('omni3d_model')`), ignore annotations
    kept with category -1, images without a usable annotation dropped when filter_empty."""
    api = Omni3D(json_file)
    model_meta = MetadataCatalog.get("omni3d_model")
    meta = MetadataCatalog.get(dataset_name)
    cats = api.loadCats(sorted(api.getCatIds(filter_settings["category_names"])))
    meta.thing_classes = [c["name"] for c in sorted(cats, key=lambda c: c["id"])]
    id_map = model_meta.thing_dataset_id_to_contiguous_id
    meta.thing_dataset_id_to_contiguous_id = id_map
    records, dropped = [], 0
    for img_id in sorted(api.imgs.keys()):
        im = api.imgs[img_id]
        rec = {"file_name": os.path.join(image_root, im["file_path"]), "dataset_id": im["dataset_id"], "height": im["height"],
               "width": im["width"], "K": im["K"], "image_id": im["id"]}
        if "p2" in im:          # KITTI projection matrix, passed along when present
            rec["p2"] = im["p2"]
        objs, any_valid = [], False
        for anno in api.imgToAnns[img_id]:
            assert anno["image_id"] == img_id
            if anno["category_id"] not in id_map and anno["category_name"] not in filter_settings["ignore_names"]:
                continue
            obj = {k: anno[k] for k in _ANN_KEYS if k in anno}
            obj["bbox_mode"] = BoxMode.XYWH_ABS
            ignore = is_ignore(anno, filter_settings, im["height"])
            obj["iscrowd"], obj["ignore"] = False, ignore
            if filter_settings["modal_2D_boxes"] and _has_box(anno, "bbox2D_tight"):
                obj["bbox"] = _xywh(anno["bbox2D_tight"])
            elif filter_settings["trunc_2D_boxes"] and _has_any(anno, "bbox2D_trunc"):
                obj["bbox"] = _xywh(anno["bbox2D_trunc"])
            elif "bbox2D_proj" in anno:
                obj["bbox"] = _xywh(anno["bbox2D_proj"])
            else:
                continue
            obj["pose"] = anno["R_cam"]
            obj["category_id"] = -1 if ignore else id_map[anno["category_id"]]
            objs.append(obj)
            any_valid = any_valid or not ignore
        if any_valid or not filter_empty:
            rec["annotations"] = objs
            records.append(rec)
        else:
            dropped += 1
    logger.info("Loaded %d images in Omni3D format from %s (%d without valid annotations filtered out)", len(records), json_file, dropped)
    return records


def simple_register(dataset_name, filter_settings=None, filter_empty=False, datasets_root_path=None, dicts=None):
    """datasets.py:125-138: `datasets/Omni3D/<name>.json`, images under `datasets/`.  Extension: `dicts` (a ready list of dataset
    dicts, e.g. omni3d_amd.synthetic.register_synthetic_dataset) registers without a file."""
    if dicts is not None:
        DatasetCatalog.register(dataset_name, lambda: dicts)
        return
    root = datasets_root_path if datasets_root_path is not None else _DEFAULT_ROOT
    path_to_json, image_root = os.path.join(root, dataset_name + ".json"), "datasets"
    DatasetCatalog.register(dataset_name, lambda: load_omni3d_json(path_to_json, image_root, dataset_name, filter_settings, filter_empty=filter_empty))
    MetadataCatalog.get(dataset_name).set(json_file=path_to_json, image_root=image_root, evaluator_type="coco")


def register_and_store_model_metadata(datasets, output_dir, filter_settings=None):
    """datasets.py:294-327: the model's category table = `filter_settings['category_names']` ordered by their global ids from
    datasets/Omni3D/stats.json; cached in <output_dir>/category_meta.json (which is what an eval-only run reads back)."""
    out = os.path.join(output_dir, "category_meta.json")
    if os.path.exists(out):
        meta = util.load_json(out)
        thing_classes = meta["thing_classes"]
        id_map = {int(a): b for a, b in meta["thing_dataset_id_to_contiguous_id"].items()}
    else:
        stats = util.load_json(os.path.join(_DEFAULT_ROOT, "stats.json"))
        names = list(filter_settings["category_names"])
        ids = [stats["categories"][stats["category_names"].index(n)]["id"] for n in names]
        order = np.argsort(ids)
        thing_classes = [names[i] for i in order]
        id_map = {int(ids[i]): k for k, i in enumerate(order)}
        util.save_json(out, {"thing_classes": thing_classes, "thing_dataset_id_to_contiguous_id": id_map})
    MetadataCatalog.get("omni3d_model").thing_classes = thing_classes
    MetadataCatalog.get("omni3d_model").thing_dataset_id_to_contiguous_id = id_map


# Category names each public Omni3D evaluation split is scored on (benchmark facts; the reference keeps the same table in
# cubercnn/data/builtin.py:3-46).  Built from four outdoor / indoor building blocks instead of one literal per split.
_KITTI = ("pedestrian", "car", "cyclist", "van", "truck")
_NUSCENES = ("pedestrian", "car", "truck", "traffic cone", "barrier", "motorcycle", "bicycle", "bus", "trailer")
_OBJECTRON = ("bicycle", "books", "bottle", "camera", "cereal box", "chair", "cup", "laptop", "shoes")
_ARKIT = ("table", "bed", "sofa", "television", "refrigerator", "chair", "oven", "machine", "stove", "shelves", "sink", "cabinet", "bathtub", "toilet")
_HYPERSIM_TEST = ("books", "chair", "towel", "blinds", "window", "lamp", "shelves", "mirror", "sink", "cabinet", "bathtub", "door", "desk", "box",
                  "bookcase", "picture", "table", "counter", "bed", "night stand", "pillow", "sofa", "television", "floor mat", "curtain", "clothes",
                  "stationery", "refrigerator")
_HYPERSIM = _HYPERSIM_TEST + ("toilet",)                              # the test annotations contain no toilet
_SUNRGBD = _HYPERSIM + ("bicycle", "bottle", "cup", "laptop", "shoes", "bin", "stove", "oven", "machine")
_OMNI3D_OUT = tuple(sorted(set(_KITTI) | set(_NUSCENES)))                                           # 11
_OMNI3D_IN = tuple(sorted(set(_SUNRGBD) | set(_ARKIT)))                                             # 38
_OMNI3D = tuple(sorted(set(_OMNI3D_OUT) | set(_OMNI3D_IN) | set(_OBJECTRON)))                       # 50
_SPLIT_FAMILIES = {"SUNRGBD": _SUNRGBD, "ARKitScenes": _ARKIT, "Objectron": _OBJECTRON, "KITTI": _KITTI, "nuScenes": _NUSCENES}


def get_omni3d_categories(dataset="omni3d"):
    """-> set of category names of `dataset` ("omni3d", "omni3d_in", "omni3d_out" or "<Family>_{train,val,test}");
    ValueError for anything else, like the reference."""
    if dataset in ("omni3d", "omni3d_in", "omni3d_out"):
        cats = {"omni3d": _OMNI3D, "omni3d_in": _OMNI3D_IN, "omni3d_out": _OMNI3D_OUT}[dataset]
        assert len(cats) == {"omni3d": 50, "omni3d_in": 38, "omni3d_out": 11}[dataset]
        return set(cats)
    family, _, split = dataset.rpartition("_")
    if family == "Hypersim" and split in ("train", "val", "test"):
        return set(_HYPERSIM_TEST if split == "test" else _HYPERSIM)
    if family in _SPLIT_FAMILIES and split in ("train", "val", "test"):
        return set(_SPLIT_FAMILIES[family])
    raise ValueError("%s dataset is not registered." % (dataset))
