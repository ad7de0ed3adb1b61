#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_bnpool.py tests/test_det.py -m gpu -q 2>&1 | tail -2
for i in 1 2; do OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 200 python bench.py --workload train --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"; done
