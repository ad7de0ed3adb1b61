#!/bin/bash
# usage (on the GPU box, through gpurun): tools/profile_final.sh <tag>   -- the reduced end-of-session measurement set:
#   <tag>_kernel_stats.csv / <tag>_trace_table.txt  rocprofv3 --kernel-trace --stats of `bench.py --workload train`
#   <tag>_pmc_families.csv                          PMC passes over tools/run_families.py (tools/pmc_run.py)
#   <tag>_bench.log                                 the default `python bench.py` line (train + iou3d + cpu baselines + drop-in loop);
# the kernel table and the PMC summary are copied to the profiles/ names bench.py hashes BEFORE the bench line is taken, so the line's
# sha256 references match the files committed afterwards.
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
OMNI_BENCH_CONDITION_STEPS=0 OMNI_BENCH_WINDOWS=1 OMNI_BENCH_SKIP_STAGE_ENDS=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o train -- python $REPO/bench.py --workload train --steps 10 --warmup 3 > $OUT/${TAG}_prof.log 2>&1
cd $REPO
f=$(find $OUT/${TAG}_prof -name 'train_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv && cp $f profiles/r06_train_final_kernel_stats.csv && head -12 $f | cut -c1-150
t=$(find $OUT/${TAG}_prof -name 'train_kernel_trace.csv' | head -1)
[ -n "$t" ] && python tools/trace_table.py $t 17 200 > $OUT/${TAG}_trace_table.txt && cp $OUT/${TAG}_trace_table.txt profiles/r06_trace_table_final.txt && head -3 $OUT/${TAG}_trace_table.txt
[ -n "$t" ] && python - "$t" $OUT/${TAG}_trace_tail.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "roi_sample_kernel" in r["Kernel_Name"]]
lo = idx[-4] if len(idx) >= 4 else 0
keep = ["Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "SGPR_Count"]
keep = [k for k in keep if k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), keep); w.writeheader()
for r in rows[lo:]:
    r = {k: r[k] for k in keep}; r["Kernel_Name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]; w.writerow(r)
PY
[ -s $OUT/${TAG}_trace_tail.csv ] && python tools/trace_timeline.py $OUT/${TAG}_trace_tail.csv $OUT/${TAG}_timeline.txt && head -3 $OUT/${TAG}_timeline.txt
find $OUT/${TAG}_prof -name '*kernel_trace.csv' -delete
if [ "${SKIP_PMC:-0}" = "0" ]; then
timeout 600 python tools/pmc_run.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc_families.csv -- python $REPO/tools/run_families.py > $OUT/${TAG}_pmc.log 2>&1
tail -3 $OUT/${TAG}_pmc.log | cut -c1-200
[ -s $OUT/${TAG}_pmc_families.csv ] && cp $OUT/${TAG}_pmc_families.csv profiles/r06_pmc_families.csv
find $OUT/${TAG}_pmc -name '*kernel_trace.csv' -delete; find $OUT/${TAG}_pmc -name '*counter_collection.csv' -delete
OMNI_BENCH_SKIP_CPU=1 timeout 600 python tools/pmc_run.py $OUT/${TAG}_pmc_iou3d $OUT/${TAG}_pmc_iou3d.csv --filter "iou_box3d|box3d_validity" -- python $REPO/bench.py --workload iou3d --steps 3 --warmup 1 > $OUT/${TAG}_pmc_iou3d.log 2>&1
tail -3 $OUT/${TAG}_pmc_iou3d.log | cut -c1-200
[ -s $OUT/${TAG}_pmc_iou3d.csv ] && cp $OUT/${TAG}_pmc_iou3d.csv profiles/r06_pmc_iou3d.csv
find $OUT/${TAG}_pmc_iou3d -name '*kernel_trace.csv' -delete; find $OUT/${TAG}_pmc_iou3d -name '*counter_collection.csv' -delete
fi   # (SKIP_PMC=1: the committed profiles/r06_pmc_*.csv of this round stay -- same kernels, see profiles/README.md)
OMNI_PIPE_TIMING=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 300 python bench.py --workload train --steps 30 --warmup 5 2>&1 | grep -E "pipe timing" > $OUT/${TAG}_pipe_timing.log
timeout 1500 python bench.py > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
cp bench_detail.json $OUT/${TAG}_bench_detail.json 2>/dev/null
tail -c 600 $OUT/${TAG}_bench.log
