"""Which configuration switch makes hipGraph capture of the training step crash?  Each variant runs in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BASE = ["MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 100,
        "MODEL.RPN.POST_NMS_TOPK_TRAIN", 30, "SOLVER.BASE_LR", 0.0002]
VARIANTS = {
    "default_128": ([], 128), "small_rois_128": (BASE, 128), "small_rois_64": (BASE, 64),
    "dla46_c_64": (BASE + ["MODEL.DLA.TYPE", "dla46_c"], 64), "fpn32_64": (BASE + ["MODEL.FPN.OUT_CHANNELS", 32], 64),
    "fc64_64": (BASE + ["MODEL.ROI_BOX_HEAD.FC_DIM", 64, "MODEL.ROI_CUBE_HEAD.FC_DIM", 64], 64),
    "light_128": (BASE + ["MODEL.DLA.TYPE", "dla46_c", "MODEL.FPN.OUT_CHANNELS", 32, "MODEL.ROI_BOX_HEAD.FC_DIM", 64, "MODEL.ROI_CUBE_HEAD.FC_DIM", 64], 128),
}
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import test_autoreplay as T
    ov, size = VARIANTS[sys.argv[1]]
    model, opt, pool = T._build("cuda", ov, size)
    model._omni_auto.warm = 1
    T._loop(model, opt, pool, 3)
    torch.cuda.synchronize()
    print("OK", sys.argv[1], "replays", model._omni_auto.replays, "failed", model._omni_auto.failed)
else:
    for name in VARIANTS:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300)
        tail = (p.stdout.strip().splitlines() or [""])[-1]
        err = [ln for ln in p.stderr.splitlines() if "Error" in ln or "error" in ln or "Fatal" in ln][-2:]
        print(f"{name:16s} rc={p.returncode} {tail} {err}", flush=True)
