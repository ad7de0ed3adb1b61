#!/bin/bash
# usage (on the GPU box, through gpurun): tools/profile_round.sh <tag>
# -> gpurun_out/<tag>_bench.log (the default bench line: train + iou3d + cpu baselines + drop-in loop),
#    <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the train workload), <tag>_trace_table.txt (per (kernel, grid)),
#    <tag>_pmc_families.csv / <tag>_pmc_iou3d.csv (PMC passes: tools/pmc_run.py), <tag>_microbench.log, <tag>_sweep.log
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_bench.log
cd /tmp
OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o train -- python $REPO/bench.py --workload train --steps 10 --warmup 3 > $OUT/${TAG}_prof.log 2>&1
cd $REPO
f=$(find $OUT/${TAG}_prof -name 'train_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv && head -25 $OUT/${TAG}_kernel_stats.csv | cut -c1-170
t=$(find $OUT/${TAG}_prof -name 'train_kernel_trace.csv' | head -1)
[ -n "$t" ] && python tools/trace_table.py $t 17 120 > $OUT/${TAG}_trace_table.txt
find $OUT/${TAG}_prof -name '*kernel_trace.csv' -delete
timeout 900 python tools/pmc_run.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc_families.csv -- python $REPO/tools/run_families.py > $OUT/${TAG}_pmc.log 2>&1
tail -40 $OUT/${TAG}_pmc.log | cut -c1-300
find $OUT/${TAG}_pmc -name '*kernel_trace.csv' -delete
OMNI_BENCH_SKIP_CPU=1 timeout 600 python tools/pmc_run.py $OUT/${TAG}_pmc_iou3d $OUT/${TAG}_pmc_iou3d.csv --filter "iou_box3d|box3d_validity" -- python $REPO/bench.py --workload iou3d --steps 3 --warmup 1 > $OUT/${TAG}_pmc_iou3d.log 2>&1
tail -8 $OUT/${TAG}_pmc_iou3d.log | cut -c1-300
find $OUT/${TAG}_pmc_iou3d -name '*kernel_trace.csv' -delete
timeout 400 python tools/sweep_batched_gemm.py > $OUT/${TAG}_sweep.log 2>&1; tail -22 $OUT/${TAG}_sweep.log
timeout 200 python tools/bench_iou3d.py > $OUT/${TAG}_iou3d_variants.log 2>&1; cat $OUT/${TAG}_iou3d_variants.log
timeout 200 python tools/winograd_error.py > $OUT/${TAG}_winograd_error.log 2>&1; tail -4 $OUT/${TAG}_winograd_error.log
