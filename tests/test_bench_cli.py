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
    # the 21 KB line of round 5 did not survive the driver's parser -- the line is capped, the rest is a file
    assert len(lines[0].encode()) <= 6000, len(lines[0])
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(emu_lib, tmp_path):
    import hashlib
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"],
               {"OMNI_BENCH_IMS": "1", "OMNI_BENCH_SIZE": "64", "OMNI_BENCH_OVERRIDES": TINY, "OMNI_BENCH_WINDOWS": "3",
                "OMNI_BENCH_DETAIL_DIR": str(tmp_path)})
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["launch"]["ranks_observed"] == 2 and res["launch"]["backend"] == "gloo"
    assert res["config"]["global_batch"] == 2 and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 1 * 1e3 / res["ms_per_step"]) < 1e-3 * res["value"] and res["value"] > 0       # whole-job rate
    # the value is the MEDIAN of the back-to-back windows, and the line says so
    w = res["windows"]
    assert len(w["ms_per_step"]) == 3 and w["min"] <= w["median"] <= w["max"] and abs(w["median"] - res["ms_per_step"]) < 1e-3 * w["median"]
    assert "median" in w["value_is"]
    assert all(x == x and abs(x) < 1e6 for x in res["loss_first_last"]) and res["skipped_steps"] == 0
    assert "functional_check_only" in res and res["nonstandard"]["device"] == "cpu"
    # both halves of BASELINE.json's metric in the default line
    io = res["iou3d"]
    assert io["unit"] == "pairs/s" and io["value"] > 0 and io["roofline"]["kernel_ms"] > 0 and "256" in io["workload"]
    for k in ("roofline", "cpu_baseline"):
        assert k in res
    # everything else went to the detail file, which the line names by hash
    d = res["detail"]
    text = open(d["file"]).read()
    assert hashlib.sha256(text.encode()).hexdigest()[:16] == d["sha256_16"]
    full = json.loads(text)
    assert len(full["launch"]["devices"]) == 2 and full["iou3d"]["config"]["pairs_per_gpu"] == 256 and full["iou3d"]["n_gpus"] == 2
    assert full["ms_per_step"] == w["median"] or abs(full["ms_per_step"] - w["median"]) < 1e-3 * w["median"]


def test_line_cap_holds_for_the_full_size_detail():
    """the cap is enforced by construction: feed compact_line the largest detail object a run can produce (17 families, N > 1 exchange)"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    fam = {"family": "x" * 200, "kernel": "gemm_nt_pf_kernel<4>" + "y" * 100, "shape": "z" * 100, "gflop": 1.0, "kernel_ms": 0.1, "tflops": 1.0,
           "frac": 0.5, "in_step": {"frac": 0.4, "avg_us": 1.0}, "pmc_source": "s" * 200}
    res = {"metric": "images/sec train DLA34_FPN b=4/GPU", "value": 373.123456789, "unit": "images/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 10.7123456, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w" * 300, "global_batch": 32, "image": "512x512", "parallelism": "dp8"},
           "windows": {"ms_per_step": [10.1234567] * 5, "min": 10.0, "median": 10.1, "max": 10.2, "value_is": "median " * 20,
                       "conditioning_steps_before_warmup": 150},
           "gpu_state": {"idle": {"sclk_mhz": 100.0, "power_w": 240.0}, "under_load": {"sclk_mhz": 2100.0, "power_w": 1000.0},
                         "under_load_all_windows": [{}] * 5, "neighbours": {"other_gpus": 7}},
           "stage_ends": {"M_end_ms": [1.0] * 7, "W_end_ms": [1.0] * 7, "window_ms_per_step": 10.9},
           "launch_mode": "m" * 500, "roofline": {"bound": "mfma", "kernel": "k" * 400, "achieved": 109.4, "peak": 157.3, "unit": "TFLOP/s",
                                                  "frac": 0.7, "traffic": 3.1e8, "families": [fam] * 17, "traffic_source": "t" * 400},
           "cpu_baseline": {"value": 0.7, "unit": "images/s", "cores": 32, "kind": "port", "sample": "s" * 900, "host": {"model": "EPYC", "physical_cores": 128}},
           "iou3d": {"metric": "IoU3D", "value": 1.95e8, "unit": "pairs/s", "ms_per_step": 0.5, "config": {"workload": "iou3d " * 30},
                     "roofline": {"bound": "hbm", "kernel": "iou_box3d_kernel", "achieved": 40.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.005,
                                  "note": "n" * 800, "valu": {"a": 1}}, "cpu_baseline": {"value": 4.5e4, "cores": 1, "kind": "port", "sample": "s" * 500,
                                                                                          "openmp": {"value": 6e5, "cores": 64}}},
           "infer": {"value": 654.0, "ms_per_step": 6.1, "config": {"x": "y" * 500}}, "resnet34": {"value": 370.0, "ms_per_step": 10.8},
           "dropin_loop_ms_per_step": 11.4, "dropin_loop_multiscale_stream": {"iterations": 320, "ms_per_iteration_whole_region": 90.0, "stats": {"a": "b" * 900}},
           "exchange": {"exposed_ms": 0.5, "all_reduce_calls_per_step": 18, "bytes_per_step": 1.9e8, "stage_timeline": {"M": [1.0] * 50}, "env": {"A": "B" * 900}},
           "launch": {"backend": "rccl", "device": "cuda", "ranks_observed": 8, "devices": ["cuda:%d" % i for i in range(8)], "one_gpu_per_rank": True}}
    line = bench.compact_line(res, {"file": "bench_detail.json", "sha256_16": "0" * 16, "bytes": 30000})
    text = json.dumps(line)
    assert len(text) <= bench.LINE_CAP == 6000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "iou3d", "windows",
              "gpu_state", "stage_ends", "detail"):
        assert k in line, k
    assert line["roofline"]["frac"] == 0.7 and line["roofline"]["families_in_detail"] == 17 and "families" not in line["roofline"]


def test_bench_refuses_a_mismatched_launcher():
    """--gpus 2 inside a 1-rank job must not print `n_gpus: 2` (nor silently measure one GPU)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", OMNI_BENCH_DEVICE="cpu")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "iou3d"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "started 1 rank" in (p.stderr + p.stdout)
