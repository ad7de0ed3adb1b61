#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/debug/capture_matrix.py > $OUT/r03i_capture_matrix.log 2>&1; cat $OUT/r03i_capture_matrix.log
timeout 600 python -m pytest tests/test_autoreplay.py tests/test_graphed.py -m gpu -q > $OUT/r03i_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|Fatal" $OUT/r03i_tests.log | head -10
