"""Omni3D on-disk format -> registered datasets, priors and the evaluation helper (omni3d_amd/cubercnn/data/datasets.py,
cubercnn/util/util.py:compute_priors, cubercnn/evaluation Omni3DEvaluator / Omni3DEvaluationHelper): CPU-side plumbing either
side of the hot path.  Pinned to the REFERENCE's own files (cubercnn/data/datasets.py, util/math_util.py) run under
oracle/ref_harness.py on the same synthetic annotation files."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

REF = "/root/reference"
KITTI = ["pedestrian", "car", "cyclist", "van", "truck"]
IDS = [31, 3, 20, 12, 7]                       # deliberately unordered: the model table must be ordered by id


def _cfg(names):
    from omni3d_amd.cubercnn.config import get_cfg_defaults
    from omni3d_amd.d2.config import get_cfg
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cubercnn_DLA34_FPN.yaml"))
    cfg.merge_from_list(["DATASETS.CATEGORY_NAMES", tuple(names), "MODEL.ROI_HEADS.NUM_CLASSES", len(names)])
    return cfg


def _write(tmp_path):
    from omni3d_amd import synthetic
    root = str(tmp_path)
    synthetic.write_omni3d_stats(root, KITTI + ["bus"], IDS + [40])
    train = synthetic.write_omni3d_dataset(root, "KITTI_train", KITTI, IDS, num_images=6, height=96, width=128, num_gt=5, seed=3, dataset_id=2)
    test = synthetic.write_omni3d_dataset(root, "KITTI_test", KITTI, IDS, num_images=4, height=96, width=128, num_gt=4, seed=4, dataset_id=2)
    # make the filter bite: one truncated, one invisible, one behind the camera, one without any 2D box, one of another category
    d = json.load(open(train))
    a = d["annotations"]
    a[0]["truncation"], a[1]["visibility"], a[2]["behind_camera"] = 0.995, 0.0, True
    a[3]["bbox2D_proj"] = a[3]["bbox2D_trunc"] = [-1, -1, -1, -1]
    a[4]["category_name"], a[4]["category_id"] = "bus", 40
    d["categories"].append({"id": 40, "name": "bus", "supercategory": "object"})
    json.dump(d, open(train, "w"))
    return root, train, test


@pytest.fixture
def in_tmp_cwd(tmp_path):
    cwd = os.getcwd()
    os.chdir(tmp_path)
    yield tmp_path
    os.chdir(cwd)


def _reset_catalogs():
    from omni3d_amd.d2.data import DatasetCatalog, MetadataCatalog
    for n in ("KITTI_train", "KITTI_test"):
        if n in DatasetCatalog:
            DatasetCatalog.remove(n)
        MetadataCatalog.pop(n, None)
    MetadataCatalog.pop("omni3d_model", None)


def test_annotation_index_filter_and_dataset_dicts(in_tmp_cwd):
    from omni3d_amd.cubercnn import data, util
    from omni3d_amd.d2.data import DatasetCatalog, MetadataCatalog
    _reset_catalogs()
    root, train, test = _write(in_tmp_cwd)
    cfg = _cfg(KITTI)
    fs = data.get_filter_settings_from_cfg(cfg)
    ds = data.Omni3D([train], filter_settings=fs)
    assert [c["name"] for c in ds.dataset["categories"]] == [n for _, n in sorted(zip(IDS, KITTI))]
    anns = ds.loadAnns(ds.getAnnIds())
    assert len(anns) == 6 * 5 - 2                                  # no 2D box -> dropped, other category -> dropped
    assert sum(a["ignore"] for a in anns) == 3 and all(len(a["bbox"]) == 4 and a["area"] == a["bbox"][2] * a["bbox"][3] for a in anns)
    assert ds.dataset["info"]["known_category_ids"] == IDS + [40]
    data.register_and_store_model_metadata(ds, str(in_tmp_cwd), fs)
    meta = MetadataCatalog.get("omni3d_model")
    assert meta.thing_classes == [n for _, n in sorted(zip(IDS, KITTI))] and meta.thing_dataset_id_to_contiguous_id == {i: k for k, i in enumerate(sorted(IDS))}
    assert os.path.exists(os.path.join(str(in_tmp_cwd), "category_meta.json"))
    data.simple_register("KITTI_train", fs, filter_empty=True)
    dicts = DatasetCatalog.get("KITTI_train")
    # the first image keeps only ignore annotations (3 filtered, 1 without a box, 1 of another category): dropped by filter_empty
    assert len(dicts) == 5 and MetadataCatalog.get("KITTI_train").json_file.endswith("KITTI_train.json")
    kept_all = data.load_omni3d_json(train, "datasets", "KITTI_train", fs, filter_empty=False)
    cats = [o["category_id"] for r in kept_all for o in r["annotations"]]
    assert len(kept_all) == 6 and min(cats) == -1 and max(cats) == 4
    assert sum(c == -1 for c in cats) == 4       # the loader keeps the box-less annotation (height 0 <= threshold -> ignore), like the reference
    assert all(os.path.exists(r["file_name"]) for r in dicts)
    priors = util.compute_priors(cfg, ds)
    assert len(priors["priors_dims_per_cat"]) == 5 and np.isfinite(np.asarray(priors["priors_dims_per_cat"])[:, 0]).all()
    # the mapper reads the files the loader points at
    from omni3d_amd.cubercnn.data import DatasetMapper3D
    out = DatasetMapper3D(cfg, is_train=False)(copy.deepcopy(dicts[0]))
    assert out["image"].shape[0] == 3 and out["image"].dtype == torch.uint8


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container only)")
def test_dataset_plumbing_matches_the_reference_files(in_tmp_cwd):
    """Same annotation files through the reference's own `Omni3D`, `load_omni3d_json`, `is_ignore` and `compute_priors`"""
    from oracle import ref_harness as H
    H.install()
    import cubercnn.data.datasets as RD                      # /root/reference/cubercnn/data/datasets.py
    from cubercnn.util import math_util as RM
    from omni3d_amd.cubercnn import data, util
    from omni3d_amd.d2.data import MetadataCatalog
    _reset_catalogs()
    root, train, test = _write(in_tmp_cwd)
    cfg = _cfg(KITTI)
    for modal, trunc in ((False, False), (True, False), (False, True)):
        fs_a, fs_b = data.get_filter_settings_from_cfg(cfg), RD.get_filter_settings_from_cfg(cfg)
        fs_a["modal_2D_boxes"] = fs_b["modal_2D_boxes"] = modal
        fs_a["trunc_2D_boxes"] = fs_b["trunc_2D_boxes"] = trunc
        assert fs_a == fs_b
        mine, ref = data.Omni3D([train, test], filter_settings=fs_a), RD.Omni3D([train, test], filter_settings=fs_b)
        assert mine.dataset["categories"] == ref.dataset["categories"] and mine.dataset["info"] == ref.dataset["info"]
        assert mine.dataset["annotations"] == ref.dataset["annotations"]
        assert mine.getAnnIds() == ref.getAnnIds() and sorted(mine.getImgIds()) == sorted(ref.getImgIds())
        some = mine.getAnnIds(imgIds=sorted(mine.imgs)[:2], catIds=[3, 12])
        assert some == ref.getAnnIds(imgIds=sorted(ref.imgs)[:2], catIds=[3, 12])
    # without category names the index fills them in (side effect the training script relies on)
    fs_a, fs_b = data.get_filter_settings_from_cfg(_cfg([])), RD.get_filter_settings_from_cfg(_cfg([]))
    fs_a["category_names"], fs_b["category_names"] = [], []
    mine, ref = data.Omni3D([train], filter_settings=fs_a), RD.Omni3D([train], filter_settings=fs_b)
    assert fs_a["category_names"] == fs_b["category_names"] and len(fs_a["category_names"]) == 6
    assert mine.dataset["annotations"] == ref.dataset["annotations"]
    # dataset dicts + priors with the model table in place
    fs = data.get_filter_settings_from_cfg(cfg)
    ds = data.Omni3D([train], filter_settings=fs)
    data.register_and_store_model_metadata(ds, str(in_tmp_cwd), fs)
    for filt in (True, False):
        a = data.load_omni3d_json(train, "datasets", "KITTI_train", fs, filter_empty=filt)
        b = RD.load_omni3d_json(train, "datasets", "KITTI_train", copy.deepcopy(fs), filter_empty=filt)
        assert a == b
    pa = util.compute_priors(cfg, ds)
    pb = RM.compute_priors(cfg, RD.Omni3D([train], filter_settings=copy.deepcopy(fs)))
    assert set(pa) == set(pb)
    for k in pb:
        np.testing.assert_allclose(np.asarray(pa[k], dtype=np.float64), np.asarray(pb[k], dtype=np.float64), rtol=1e-12, atol=0, equal_nan=True, err_msg=k)
    assert MetadataCatalog.get("KITTI_train").thing_classes == sorted(KITTI, key=lambda n: IDS[KITTI.index(n)])
    # CLUSTER_BINS > 1: the 1-D k-means over the 2D scale and the per-cluster depth statistics (math_util.py:401-485)
    from omni3d_amd import synthetic
    big = synthetic.write_omni3d_dataset(root, "KITTI_big", KITTI[:4], IDS[:4], num_images=24, height=96, width=128, num_gt=5, seed=9, dataset_id=2)
    populated = 0
    for nbins in (2, 3, 5):           # the fifth category has no samples: the anchor-range placeholders
        cfg.defrost()
        cfg.MODEL.ROI_CUBE_HEAD.CLUSTER_BINS = nbins
        pa = util.compute_priors(cfg, data.Omni3D([big], filter_settings=copy.deepcopy(fs)))
        pb = RM.compute_priors(cfg, RD.Omni3D([big], filter_settings=copy.deepcopy(fs)))
        assert len(pa["priors_bins"]) == len(pb["priors_bins"]) == len(KITTI)
        for (na, sa, za), (nb, sb, zb) in zip(pa["priors_bins"], pb["priors_bins"]):
            assert na == nb and len(sa) == len(sb) == nbins
            np.testing.assert_allclose(sa, sb, rtol=1e-6)
            np.testing.assert_allclose(np.asarray(za, dtype=np.float64), np.asarray(zb, dtype=np.float64), rtol=1e-9, equal_nan=True)
            populated += int(len({round(v, 3) for v in sa}) == nbins and sa[0] != 32)
        for k in pb:
            if k != "priors_bins":
                np.testing.assert_allclose(np.asarray(pa[k], dtype=np.float64), np.asarray(pb[k], dtype=np.float64), rtol=1e-12, equal_nan=True)
    assert populated > 0, "no category had enough samples to exercise the clustering itself"


def test_evaluation_helper_scores_perfect_predictions(emu_lib, in_tmp_cwd):
    """ground truth fed back as predictions: every present category scores 100 in 2D and 3D, the per-split and the <Concat>
    tables agree, the Omni3D aggregates are NaN (categories missing), files land where tools/train_net.py:do_test reads them"""
    from omni3d_amd.cubercnn import data
    from omni3d_amd.cubercnn.evaluation import Omni3DEvaluationHelper
    from omni3d_amd.d2.data import MetadataCatalog
    _reset_catalogs()
    root, train, test = _write(in_tmp_cwd)
    cfg = _cfg(KITTI)
    fs = data.get_filter_settings_from_cfg(cfg)
    ds = data.Omni3D([train], filter_settings=fs)
    data.register_and_store_model_metadata(ds, str(in_tmp_cwd), fs)
    fs_test = data.get_filter_settings_from_cfg(cfg)
    fs_test.update(visibility_thres=cfg.TEST.VISIBILITY_THRES, truncation_thres=cfg.TEST.TRUNCATION_THRES, min_height_thres=0.0625, max_depth=1e8)
    data.simple_register("KITTI_test", fs_test, filter_empty=False)
    helper = Omni3DEvaluationHelper(["KITTI_test"], fs_test, os.path.join(str(in_tmp_cwd), "inference"), iter_label="7")
    gt = data.Omni3D([test], filter_settings=copy.deepcopy(fs_test))
    id_map = MetadataCatalog.get("omni3d_model").thing_dataset_id_to_contiguous_id
    preds = []
    for img_id, im in sorted(gt.imgs.items()):
        recs = [{"image_id": img_id, "category_id": id_map[a["category_id"]], "bbox": list(a["bbox"]), "score": 0.9 - 0.01 * k,
                 "depth": a["depth"], "bbox3D": a["bbox3D"]} for k, a in enumerate(gt.imgToAnns[img_id]) if not a["ignore"]]
        preds.append({"image_id": img_id, "K": im["K"], "width": im["width"], "height": im["height"], "instances": recs})
    helper.add_predictions("KITTI_test", preds)
    helper.save_predictions("KITTI_test")
    helper.evaluate("KITTI_test")
    analysis, omni = helper.summarize_all()
    assert os.path.exists(os.path.join(str(in_tmp_cwd), "inference", "KITTI_test", "instances_predictions.pth"))
    assert os.path.exists(os.path.join(str(in_tmp_cwd), "inference", "KITTI_test", "omni_instances_results.json"))
    for name in ("KITTI_test", "<Concat>"):
        assert abs(analysis[name]["AP2D"] - 100.0) < 1e-6 and abs(analysis[name]["AP3D"] - 100.0) < 1e-6, analysis[name]
        assert analysis[name]["iters"] == "7"
    assert abs(omni["KITTI_test"]["AP3D"] - 100.0) < 1e-6                  # the split's own category list is complete
    assert np.isnan(omni["Omni3D"]["AP2D"]) and np.isnan(omni["Omni3D_Out"]["AP3D"])


def test_dataset_balancing_weights():
    """DATALOADER.BALANCE_DATASETS (cubercnn/data/build.py:66-121): (1 - share of the source) / smallest such value"""
    from omni3d_amd.cubercnn.data.build import dataset_balance_weights
    dicts = [{"dataset_id": 0}] * 6 + [{"dataset_id": 1}] * 2 + [{"dataset_id": 2}] * 2
    w = dataset_balance_weights(dicts, {0: "A", 1: "B", 2: "B"})
    assert torch.allclose(w, torch.tensor([1.0] * 6 + [1.5] * 4))
    assert torch.equal(dataset_balance_weights(dicts, {0: "A", 1: "A", 2: "A"}), torch.ones(10))
