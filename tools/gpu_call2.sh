#!/bin/bash
# round 3, call 2: A/B of the deep-prefetch tile GEMMs, IoU3D lane widths, current per-launch table, new parity tests
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv.py tests/test_iou3d.py -m gpu -q -x > $OUT/r03b_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r03b_tests.log
timeout 300 python tools/sweep_batched_gemm.py > $OUT/r03b_sweep.log 2>&1; cat $OUT/r03b_sweep.log
timeout 200 python tools/bench_iou3d.py > $OUT/r03b_iou3d_widths.log 2>&1; cat $OUT/r03b_iou3d_widths.log
timeout 300 python tools/layer_table.py > $OUT/r03b_layer_table.txt 2> $OUT/r03b_layer_table.err; head -75 $OUT/r03b_layer_table.txt | cut -c1-150
