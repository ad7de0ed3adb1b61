"""Dataset-side helpers the training script imports from `cubercnn.data` (tools/train_net.py:33-40, :338-387).  JSON
parsing / filtering / category bookkeeping is CPU-side data plumbing outside the MI355X hot path: the
registration hooks are real (so DatasetCatalog-driven code runs), the Omni3D JSON reader is not shipped."""
from ...d2.data import DatasetCatalog, MetadataCatalog


def get_filter_settings_from_cfg(cfg=None):
    """datasets.py: the filter dictionary built from cfg.DATASETS / cfg.TEST keys"""
    if cfg is None:
        return {"category_names": [], "ignore_names": [], "truncation_thres": 0.99, "visibility_thres": 0.01, "min_height_thres": 0.00,
                "max_height_thres": 1.50, "modal_2D_boxes": False, "trunc_2D_boxes": False, "max_depth": 1e8}
    d = cfg.DATASETS
    return {"category_names": d.CATEGORY_NAMES, "ignore_names": d.IGNORE_NAMES, "truncation_thres": d.TRUNCATION_THRES,
            "visibility_thres": d.VISIBILITY_THRES, "min_height_thres": d.MIN_HEIGHT_THRES, "modal_2D_boxes": d.MODAL_2D_BOXES,
            "trunc_2D_boxes": d.TRUNC_2D_BOXES, "max_depth": d.MAX_DEPTH, "max_height_thres": 1.50}


def simple_register(dataset_name, filter_settings=None, filter_empty=False, datasets_root_path=None, dicts=None):
    """registers `dataset_name`; `dicts` (list of dataset dicts) stands in for the JSON file of the reference"""
    if dicts is None:
        raise NotImplementedError("reading datasets/Omni3D/<name>.json is outside the MI355X hot path; pass dicts=... "
                                  "(omni3d_amd.synthetic.register_synthetic_dataset does)")
    DatasetCatalog.register(dataset_name, lambda: dicts)
    MetadataCatalog.get(dataset_name).set = None


def load_omni3d_json(*args, **kwargs):
    raise NotImplementedError("Omni3D JSON parsing (cubercnn/data/datasets.py:170-330) is CPU-side dataset plumbing, out of the hot-path scope")


def get_omni3d_categories(dataset="omni3d"):
    raise NotImplementedError("category tables (cubercnn/data/builtin.py) are dataset plumbing, out of the hot-path scope")


def register_and_store_model_metadata(datasets, output_dir, filter_settings=None):
    raise NotImplementedError("model metadata bookkeeping (datasets.py:392-448) is dataset plumbing, out of the hot-path scope")


class Omni3D:
    def __init__(self, *a, **k):
        raise NotImplementedError("the COCO-style Omni3D annotation index (datasets.py:18-167) is dataset plumbing, out of the hot-path scope")
