#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_iou3d.py -m gpu -q > $OUT/r03g_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed" $OUT/r03g_tests.log | head -10
timeout 900 python tools/debug/capture_matrix.py > $OUT/r03g_capture_matrix.log 2>&1; cat $OUT/r03g_capture_matrix.log
