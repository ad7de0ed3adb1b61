#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/_gpu_ab.sh "-" "OMNI_FC_WGRAD_SPLITS=1" "OMNI_FC_WGRAD_WGS=224" "OMNI_FC_WGRAD_WGS=192" "OMNI_FC_WGRAD_WGS=128" "OMNI_FC_BALANCED=0" "OMNI_FC_WGRAD_ENGINE_MIN_ROWS=100000000" 2>&1 | tee $OUT/r05b_ab.log
OMNI_FC_WGRAD_WGS=192 bash tools/_gpu_prof.sh r05b_wgs192 2>&1 | tail -2
python tools/trace_timeline.py $OUT/r05b_wgs192_trace_tail.csv $OUT/r05b_wgs192_timeline.txt
OMNI_FC_WGRAD_ENGINE_MIN_ROWS=100000000 bash tools/_gpu_prof.sh r05b_tile 2>&1 | tail -2
python tools/trace_timeline.py $OUT/r05b_tile_trace_tail.csv $OUT/r05b_tile_timeline.txt
