#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_iou3d.py tests/test_conv.py -m gpu -q -x > $OUT/r03c_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r03c_tests.log
timeout 200 python tools/bench_iou3d.py > $OUT/r03c_iou3d_variants.log 2>&1; cat $OUT/r03c_iou3d_variants.log
timeout 200 python tools/bench_wino_dy.py > $OUT/r03c_wino_dy.log 2>&1; cat $OUT/r03c_wino_dy.log
OMNI_BENCH_SKIP_CPU=1 timeout 300 python bench.py --workload train > $OUT/r03c_bench_train.log 2> $OUT/r03c_bench_train.err; echo "bench rc=$?"; python - <<'PY'
import json
for ln in open("gpurun_out/r03c_bench_train.log"):
    if ln.startswith("{"):
        d = json.loads(ln); print("images/s", d["value"], "ms", d["ms_per_step"])
        for f in d["roofline"]["families"]: print("  %-60s %7.3f ms %6.1f TF %.2f" % (f["family"][:60], f["kernel_ms"], f["tflops"], f["frac"]))
PY
cd /tmp
OMNI_BENCH_SKIP_CPU=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03c_prof -o train -- python $REPO/bench.py --workload train --steps 10 --warmup 3 > $OUT/r03c_prof.log 2>&1
cd $REPO
f=$(find $OUT/r03c_prof -name 'train_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/r03c_kernel_stats.csv && head -22 $OUT/r03c_kernel_stats.csv | cut -c1-150
find $OUT/r03c_prof -name '*kernel_trace.csv' -delete
