#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/debug/capture_matrix.py > $OUT/r03h_capture_matrix.log 2>&1; cat $OUT/r03h_capture_matrix.log
timeout 300 python -m pytest tests/test_conv.py -m gpu -q -k "linear" > $OUT/r03h_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed" $OUT/r03h_tests.log | head -10
for v in 1 0; do OMNI_FC_DGRAD_NT=$v OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 200 python bench.py --workload train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('dgrad_nt=$v %.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"; done
