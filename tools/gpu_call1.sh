#!/bin/bash
# round 3, call 1: full GPU suite + default bench line + 2-rank functional check + baseline kernel table
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/r03a_gputests.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/r03a_gputests.log
timeout 600 python bench.py > $OUT/r03a_bench.log 2> $OUT/r03a_bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/r03a_bench.log; tail -5 $OUT/r03a_bench.err
OMNI_BENCH_ONE_DEVICE=1 OMNI_BENCH_SKIP_CPU=1 timeout 600 python bench.py --gpus 2 --steps 5 > $OUT/r03a_bench_2rank.log 2> $OUT/r03a_bench_2rank.err; echo "2rank rc=$?"; tail -c 600 $OUT/r03a_bench_2rank.log; tail -5 $OUT/r03a_bench_2rank.err
cd /tmp
OMNI_BENCH_SKIP_CPU=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03a_prof -o train -- python $REPO/bench.py --workload train --steps 10 --warmup 3 > $OUT/r03a_prof.log 2>&1
cd $REPO
f=$(find $OUT/r03a_prof -name 'train_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/r03a_kernel_stats.csv && head -30 $OUT/r03a_kernel_stats.csv | cut -c1-160
find $OUT/r03a_prof -name '*kernel_trace.csv' -delete
