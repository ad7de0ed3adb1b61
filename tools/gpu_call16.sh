#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $OUT/r03p_gputests.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/r03p_gputests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r03p_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/r03p_smoke.log | cut -c1-200
bash tools/profile_round.sh r03p
OMNI_BENCH_ONE_DEVICE=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/r03p_bench_2rank.log 2> $OUT/r03p_bench_2rank.err; echo "2rank rc=$?"; python - <<'PY'
import json
for ln in open("gpurun_out/r03p_bench_2rank.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print("2-rank:", d["n_gpus"], d["launch"], d["value"], d.get("iou3d",{}).get("n_gpus"))
PY
