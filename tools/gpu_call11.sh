#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_autoreplay.py "tests/test_model_parity.py::test_training_step_matches_reference_gpu" tests/test_conv.py -m gpu -q > $OUT/r03k_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed" $OUT/r03k_tests.log | head -10 | cut -c1-300
timeout 400 python tools/sweep_batched_gemm.py > $OUT/r03k_sweep.log 2>&1; tail -22 $OUT/r03k_sweep.log
for i in 1 2; do OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 200 python bench.py --workload train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"; done
timeout 900 python tools/pmc_run.py $OUT/r03k_pmc $OUT/r03k_pmc_families.csv --filter "gemm_" -- python $REPO/tools/run_families.py > $OUT/r03k_pmc.log 2>&1
find $OUT/r03k_pmc -name '*kernel_trace.csv' -delete
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r03k_pmc_families.csv")):
    busy=float(r["SQ_VALU_MFMA_BUSY_CYCLES"])/(float(r["GRBM_GUI_ACTIVE"])/8*1024)
    print("%-34s grid %8s n=%2s fetchx2 %7s MB write %7s MB mfma_busy %.2f" % (r["kernel"][:34], r["grid"], r["launches"], r["FETCH_SIZE_x2_MB"], r["WRITE_SIZE_MB"], busy))
PY
