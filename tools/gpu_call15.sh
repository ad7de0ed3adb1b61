#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/bench_fc_wgrad.py 2>&1 | tee $OUT/r03o_fc_wgrad.log
