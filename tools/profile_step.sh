#!/bin/bash
# usage: tools/profile_step.sh <tag>  -> gpurun_out/<tag>_trace_table.txt, <tag>_trace_tail.csv (last 3 steps), <tag>_pipe.log
set -u
TAG=${1:-r04x}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
OMNI_PIPE_TIMING=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 300 python bench.py --workload train --steps 30 --warmup 5 2>&1 | grep -E "pipe timing|images/sec" | cut -c1-400 > $OUT/${TAG}_pipe.log; cat $OUT/${TAG}_pipe.log | cut -c1-330
cd /tmp
OMNI_BENCH_CONDITION_STEPS=0 OMNI_BENCH_WINDOWS=1 OMNI_BENCH_SKIP_STAGE_ENDS=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o train -- python $REPO/bench.py --workload train --steps 10 --warmup 3 > $OUT/${TAG}_prof.log 2>&1
cd $REPO
t=$(find $OUT/${TAG}_prof -name 'train_kernel_trace.csv' | head -1)
[ -n "$t" ] && python tools/trace_table.py $t 17 200 > $OUT/${TAG}_trace_table.txt
f=$(find $OUT/${TAG}_prof -name 'train_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv
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
rm -rf $OUT/${TAG}_prof
head -1 $OUT/${TAG}_trace_table.txt
