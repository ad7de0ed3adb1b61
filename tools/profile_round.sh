#!/bin/bash
# usage (on the GPU box, through gpurun): tools/profile_round.sh <tag> [steps]
# -> gpurun_out/<tag>_bench.log, <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py),
#    <tag>_pmc_families.csv (PMC passes on tools/run_families.py), <tag>_microbench.log
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
tail -c 3000 $OUT/${TAG}_bench.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o train -- python $REPO/bench.py --steps 10 --warmup 3 > $OUT/${TAG}_prof.log 2>&1
cd $REPO
f=$(find $OUT/${TAG}_prof -name 'train_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv && head -40 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
find $OUT/${TAG}_prof -name '*kernel_trace.csv' -delete
timeout 900 python tools/pmc_run.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc_families.csv -- python $REPO/tools/run_families.py > $OUT/${TAG}_pmc.log 2>&1
tail -60 $OUT/${TAG}_pmc.log | cut -c1-400
find $OUT/${TAG}_pmc -name '*kernel_trace.csv' -delete
timeout 600 python tools/pmc_run.py $OUT/${TAG}_pmc_iou3d $OUT/${TAG}_pmc_iou3d.csv --filter "iou_box3d|box3d_validity" -- python $REPO/bench.py --workload iou3d --steps 3 --warmup 1 > $OUT/${TAG}_pmc_iou3d.log 2>&1
tail -8 $OUT/${TAG}_pmc_iou3d.log | cut -c1-400
find $OUT/${TAG}_pmc_iou3d -name '*kernel_trace.csv' -delete
timeout 600 python tools/bench_kernels.py > $OUT/${TAG}_microbench.log 2>&1
tail -40 $OUT/${TAG}_microbench.log
python bench.py --workload iou3d > $OUT/${TAG}_bench_iou3d.log 2>/dev/null; tail -c 2500 $OUT/${TAG}_bench_iou3d.log
