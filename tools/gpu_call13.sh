#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
ab() { name=$1; shift
  v=$(env "$@" OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 200 python bench.py --workload train --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%.2f images/s  %.3f ms  %s' % (d['value'], d['ms_per_step'], d['launch_mode'][:40]))")
  echo "$name: $v" | tee -a $OUT/r03m_ab.log; }
: > $OUT/r03m_ab.log
ab default A=1
ab cuts_none OMNI_PIPE_CUTS=
ab cuts_p2 OMNI_PIPE_CUTS=p2
ab cuts_p2p3 OMNI_PIPE_CUTS=p2,p3
ab cuts_p2p3p4 OMNI_PIPE_CUTS=p2,p3,p4
ab cuts_all OMNI_PIPE_CUTS=p2,p3,p4,p5
ab wgrad_batch4 OMNI_WGRAD_BATCH=4
ab default_again A=1
timeout 300 python -m pytest tests/test_model_parity.py -m gpu -q > $OUT/r03m_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed" $OUT/r03m_tests.log | head -5 | cut -c1-300
