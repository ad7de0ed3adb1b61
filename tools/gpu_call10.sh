#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $OUT/r03j_gputests.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/r03j_gputests.log | cut -c1-200
bash tools/profile_round.sh r03j
