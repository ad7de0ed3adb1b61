#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_iou3d.py tests/test_evaluator.py tests/test_eval_match.py -m gpu -q -x > $OUT/r03d_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r03d_tests.log
timeout 200 python tools/bench_iou3d.py > $OUT/r03d_iou3d_variants.log 2>&1; cat $OUT/r03d_iou3d_variants.log
ab() { # name, env...
  name=$1; shift
  v=$(env "$@" OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 200 python bench.py --workload train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))")
  echo "$name: $v" | tee -a $OUT/r03d_ab.log
}
: > $OUT/r03d_ab.log
ab base A=1
ab dy_split_4MB OMNI_WINO_DY_SPLIT=4000000
ab dy_split_1MB OMNI_WINO_DY_SPLIT=1000000
ab min_tiles_64 OMNI_WINO_MIN_TILES=64
ab dgrad_tiles_256 OMNI_WINO_DGRAD_MIN_TILES=256
ab dgrad_tiles_64_min64 OMNI_WINO_DGRAD_MIN_TILES=64 OMNI_WINO_MIN_TILES=64
ab all OMNI_WINO_DY_SPLIT=4000000 OMNI_WINO_DGRAD_MIN_TILES=256 OMNI_WINO_MIN_TILES=64
ab base_again A=1
cd /tmp
OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/r03d_prof -o train -- python $REPO/bench.py --workload train --steps 10 --warmup 3 > $OUT/r03d_prof.log 2>&1
cd $REPO
f=$(find $OUT/r03d_prof -name 'train_kernel_trace.csv' | head -1)
[ -n "$f" ] && python tools/trace_table.py $f 16 90 > $OUT/r03d_trace_table.txt && head -95 $OUT/r03d_trace_table.txt
find $OUT/r03d_prof -name '*kernel_trace.csv' -delete
