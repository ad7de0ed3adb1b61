#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv.py -m gpu -q -k "stem" > $OUT/r03n_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed" $OUT/r03n_tests.log | head -5
timeout 300 python tools/bench_kernels.py 2>&1 | grep -A4 "direct stem" | tee $OUT/r03n_stem.log
for i in 1 2; do OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 200 python bench.py --workload train --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"; done
