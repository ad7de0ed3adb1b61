"""Bodies of tests/test_reference_train_loop.py.  Each case runs in its OWN interpreter (`python tests/ref_train_loop_cases.py <case>
<tmp dir>`): `omni3d_amd.install()` makes `cubercnn` / `detectron2` resolve to this package, while other tests import the
REFERENCE's `cubercnn` through oracle/ref_harness.py -- the two must never share a `sys.modules`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

REF = "/root/reference/tools/train_net.py"


def _emulator():
    """what the `emu_lib` fixture does: kernels through the host-emulated build (tests/hipemu)"""
    from omni3d_amd import lib as L
    L._install_for_tests(L.HipLibrary(os.path.join(ROOT, "tests", "hipemu", "libomni3d_emu.so"), emulated=True))
    torch.Tensor.cuda = lambda self, *a, **k: self        # `.cuda()` is hard-coded in the script (:237, :263)


def _load_reference_script():
    import omni3d_amd
    omni3d_amd.install()
    spec = importlib.util.spec_from_file_location("reference_train_net", REF)
    mod = importlib.util.module_from_spec(spec)
    cwd = os.getcwd()
    try:
        spec.loader.exec_module(mod)          # executes the reference file's own imports against this package
    finally:
        os.chdir(cwd)
    return mod



def imports_resolve(tmp_path):
    mod = _load_reference_script()
    for name in ("do_train", "do_test", "setup", "main", "allreduce_dict"):
        assert hasattr(mod, name)
    import omni3d_amd.cubercnn.solver as S
    assert mod.build_optimizer is S.build_optimizer            # the script's names ARE this package's objects



def do_train(tmp_path):
    _emulator()
    mod = _load_reference_script()
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.config import get_cfg_defaults
    from omni3d_amd.d2.config import get_cfg
    priors = synthetic.make_priors(50)
    synthetic.register_synthetic_dataset("synthetic_train", num_images=4, height=64, width=64, num_gt=3, seed=7, priors=priors)
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "cubercnn_DLA34_FPN.yaml"))
    iters = int(os.environ.get("OMNI_DO_TRAIN_ITERS", "2"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "MODEL.WEIGHTS", "synthetic://random-init", "MODEL.WEIGHTS_PRETRAIN", "", "OUTPUT_DIR", str(tmp_path),
                         "DATASETS.TRAIN", ("synthetic_train",), "SOLVER.IMS_PER_BATCH", 1, "SOLVER.MAX_ITER", iters, "SOLVER.BASE_LR", 0.001,
                         "SOLVER.STEPS", (), "SOLVER.WARMUP_ITERS", 1, "SOLVER.CHECKPOINT_PERIOD", 1, "TEST.EVAL_PERIOD", 0,
                         "INPUT.MIN_SIZE_TRAIN", (64,), "INPUT.MAX_SIZE_TRAIN", 64, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16,
                         "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 100, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 30])
    torch.manual_seed(0)
    model = mod.build_model(cfg, priors=priors)
    before = {k: v.clone() for k, v in model.state_dict().items() if "fc2.weight" in k}
    ok = mod.do_train(cfg, model, dataset_id_to_unknown_cats={0: set()}, dataset_id_to_src={0: "synthetic"}, resume=False)
    assert ok is True                                                           # ran to max_iter without a restart request
    after = model.state_dict()
    assert any(float((after[k] - v).abs().max()) > 0 for k, v in before.items())         # the optimizer stepped
    files = sorted(os.listdir(tmp_path))
    assert "model_final.pth" in files and "model_recent.pth" in files and "last_checkpoint" in files       # PeriodicCheckpointerOnlyOne
    ck = torch.load(os.path.join(tmp_path, "model_final.pth"), weights_only=False)
    assert {"model", "optimizer", "scheduler", "iteration"} <= set(ck) and ck["iteration"] == iters - 1
    assert all("momentum_buffer" in s for s in ck["optimizer"]["state"].values())                        # torch-SGD format



def main_literal(tmp_path):
    """The reference's `main(args)` (tools/train_net.py:353-466), unchanged, from a working directory that holds a synthetic split
    in the Omni3D on-disk format: setup() -> simple_register -> Omni3D index -> category metadata -> compute_priors ->
    build_model -> do_train -> do_test (test loader, inference_on_dataset, Omni3DEvaluationHelper, summarize_all)."""
    _emulator()
    import argparse
    mod = _load_reference_script()
    from omni3d_amd import synthetic
    from omni3d_amd.d2.data import DatasetCatalog, MetadataCatalog
    for n in ("KITTI_train", "KITTI_test", "omni3d_model"):
        if n in DatasetCatalog:
            DatasetCatalog.remove(n)
        MetadataCatalog.pop(n, None)
    names, ids = ["pedestrian", "car", "cyclist", "van", "truck"], [31, 3, 20, 12, 7]
    root = str(tmp_path)
    synthetic.write_omni3d_stats(root, names, ids)
    synthetic.write_omni3d_dataset(root, "KITTI_train", names, ids, num_images=4, height=64, width=64, num_gt=3, seed=11, dataset_id=2)
    synthetic.write_omni3d_dataset(root, "KITTI_test", names, ids, num_images=2, height=64, width=64, num_gt=3, seed=12, dataset_id=2,
                                   image_id_base=900000)
    out = os.path.join(root, "output")
    iters = int(os.environ.get("OMNI_DO_TRAIN_ITERS", "2"))
    args = argparse.Namespace(config_file=os.path.join(ROOT, "configs", "cubercnn_DLA34_FPN.yaml"), resume=False, eval_only=False, num_gpus=1,
                              num_machines=1, machine_rank=0, dist_url="tcp://127.0.0.1:29599",
                              opts=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "MODEL.WEIGHTS", "synthetic://random-init", "MODEL.WEIGHTS_PRETRAIN", "",
                                    "OUTPUT_DIR", out, "DATASETS.TRAIN", ("KITTI_train",), "DATASETS.TEST", ("KITTI_test",),
                                    "DATASETS.CATEGORY_NAMES", tuple(names), "MODEL.ROI_HEADS.NUM_CLASSES", len(names),
                                    "SOLVER.IMS_PER_BATCH", 1, "SOLVER.MAX_ITER", iters, "SOLVER.BASE_LR", 0.001, "SOLVER.STEPS", (),
                                    "SOLVER.WARMUP_ITERS", 1, "SOLVER.CHECKPOINT_PERIOD", 1, "TEST.EVAL_PERIOD", 0,
                                    "INPUT.MIN_SIZE_TRAIN", (64,), "INPUT.MAX_SIZE_TRAIN", 64, "INPUT.MIN_SIZE_TEST", 64, "INPUT.MAX_SIZE_TEST", 64,
                                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 16,
                                    "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 100, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 30,
                                    "MODEL.RPN.PRE_NMS_TOPK_TEST", 100, "MODEL.RPN.POST_NMS_TOPK_TEST", 30])
    cwd = os.getcwd()
    os.chdir(root)
    try:
        mod.main(args)
    finally:
        os.chdir(cwd)
    files = sorted(os.listdir(out))
    assert "model_final.pth" in files and "category_meta.json" in files and "config.yaml" in files
    inf = os.path.join(out, "inference", "iter_final", "KITTI_test")
    assert os.path.exists(os.path.join(inf, "instances_predictions.pth"))
    assert MetadataCatalog.get("omni3d_model").thing_classes == [n for _, n in sorted(zip(ids, names))]
    first = torch.load(os.path.join(inf, "instances_predictions.pth"), weights_only=False)
    # ---- the script's --eval-only path (:365-379, :432-438): categories from <config dir>/category_meta.json, weights through
    # DetectionCheckpointer.resume_or_load, straight to do_test; same weights => same detections
    import shutil
    cfg_dir = os.path.join(root, "cfgdir")
    os.makedirs(cfg_dir)
    for f in os.listdir(os.path.join(ROOT, "configs")):
        shutil.copy(os.path.join(ROOT, "configs", f), cfg_dir)
    shutil.copy(os.path.join(out, "category_meta.json"), cfg_dir)
    out2 = os.path.join(root, "output_eval")
    opts = list(args.opts)
    opts[opts.index("OUTPUT_DIR") + 1] = out2
    opts[opts.index("MODEL.WEIGHTS") + 1] = os.path.join(out, "model_final.pth")
    args2 = argparse.Namespace(**{**vars(args), "config_file": os.path.join(cfg_dir, "cubercnn_DLA34_FPN.yaml"), "eval_only": True, "opts": opts})
    MetadataCatalog.pop("omni3d_model", None)
    for n in ("KITTI_train", "KITTI_test"):
        if n in DatasetCatalog:
            DatasetCatalog.remove(n)
        MetadataCatalog.pop(n, None)
    os.chdir(root)
    try:
        mod.main(args2)
    finally:
        os.chdir(cwd)
    second = torch.load(os.path.join(out2, "inference", "iter_final", "KITTI_test", "instances_predictions.pth"), weights_only=False)
    assert [len(p["instances"]) for p in first] == [len(p["instances"]) for p in second]
    for a, b in zip(first, second):
        for x, y in zip(a["instances"], b["instances"]):
            assert x["category_id"] == y["category_id"] and abs(x["score"] - y["score"]) < 1e-5



if __name__ == "__main__":
    {"imports_resolve": imports_resolve, "do_train": do_train, "main_literal": main_literal}[sys.argv[1]](sys.argv[2])
    print("CASE-OK")
