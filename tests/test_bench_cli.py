"""`python bench.py --gpus 2` end to end, the way the driver calls it (no launcher, no WORLD_SIZE): bench.py must start the two
ranks itself (reference: tools/train_net.py:500-510 `launch(main, args.num_gpus, ...)`), exchange gradients between them and
report what it observed.  GPU-less form: the ranks run the host-compiled kernels (tests/emu_site/sitecustomize.py, a test
seam the product does not know about) on a shrunken model over gloo; the line says so (`functional_check_only`)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

TINY = ("MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE 16 MODEL.RPN.BATCH_SIZE_PER_IMAGE 16 MODEL.RPN.PRE_NMS_TOPK_TRAIN 100 "
        "MODEL.RPN.POST_NMS_TOPK_TRAIN 30 MODEL.DLA.TYPE dla46_c MODEL.FPN.OUT_CHANNELS 32 MODEL.ROI_BOX_HEAD.FC_DIM 64 "
        "MODEL.ROI_CUBE_HEAD.FC_DIM 64")


def _run(args, extra_env):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests", "emu_site") + os.pathsep + os.environ.get("PYTHONPATH", ""),
               OMNI_EMULATE="1", OMNI_BENCH_DEVICE="cpu", OMNI_BENCH_SKIP_CPU="1", OMNI_BENCH_IOU_PAIRS="256", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(emu_lib):
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"],
               {"OMNI_BENCH_IMS": "1", "OMNI_BENCH_SIZE": "64", "OMNI_BENCH_OVERRIDES": TINY})
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["launch"]["ranks_observed"] == 2 and len(res["launch"]["devices"]) == 2 and res["launch"]["backend"] == "gloo"
    assert res["config"]["global_batch"] == 2 and res["scaling"] == "weak"
    assert res["value"] > 0 and abs(res["value"] - 2 * 1 * 1e3 / res["ms_per_step"]) < 1e-6 * res["value"]       # whole-job rate
    assert all(x == x and abs(x) < 1e6 for x in res["loss_first_last"]) and res["skipped_steps"] == 0
    assert "functional_check_only" in res and res["nonstandard"]["device"] == "cpu"
    # both halves of BASELINE.json's metric in the default line
    io = res["iou3d"]
    assert io["unit"] == "pairs/s" and io["n_gpus"] == 2 and io["value"] > 0 and io["roofline"]["kernel_ms"] > 0
    assert io["config"]["pairs_per_gpu"] == 256


def test_bench_refuses_a_mismatched_launcher():
    """--gpus 2 inside a 1-rank job must not print `n_gpus: 2` (nor silently measure one GPU)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", OMNI_BENCH_DEVICE="cpu")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "iou3d"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "started 1 rank" in (p.stderr + p.stdout)
