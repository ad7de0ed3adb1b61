#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_iou3d.py tests/test_autoreplay.py tests/test_graphed.py tests/test_solver.py tests/test_boundary_contracts.py -m gpu -q > $OUT/r03e_tests.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r03e_tests.log
timeout 900 python bench.py > $OUT/r03e_bench.log 2> $OUT/r03e_bench.err; echo "bench rc=$?"; tail -3 $OUT/r03e_bench.err
python - <<'PY'
import json
for ln in open("gpurun_out/r03e_bench.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("images/s", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"])
        print("dropin", d.get("dropin_loop_ms_per_step"), json.dumps(d.get("dropin_loop"))[:900])
        print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:700])
        print("iou3d", d["iou3d"]["value"], d["iou3d"]["roofline"]["kernel_ms"], json.dumps(d["iou3d"]["cpu_baseline"])[:300])
PY
