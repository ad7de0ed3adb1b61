#!/bin/bash
# round 5, GPU call 1: BatchNorm finalize folded into apply (OMNI_BN_FUSE_ROWS) and wave priority of the critical path (OMNI_LIB_SUFFIX)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bnpool.py -m gpu -x -q 2>&1 | tail -3 > $OUT/r05a_bn_tests.log; cat $OUT/r05a_bn_tests.log
bash tools/_gpu_ab.sh "OMNI_BN_FUSE_ROWS=0" "-" "OMNI_BN_FUSE_ROWS=4096" "OMNI_LIB_SUFFIX=_prio2" "OMNI_LIB_SUFFIX=_prio3" "OMNI_BN_FUSE_ROWS=256" 2>&1 | tee $OUT/r05a_ab.log
bash tools/_gpu_prof.sh r05a 2>&1 | tail -3
python tools/trace_timeline.py $OUT/r05a_trace_tail.csv $OUT/r05a_timeline.txt
OMNI_LIB_SUFFIX=_prio2 bash tools/_gpu_prof.sh r05a_prio2 2>&1 | tail -3
python tools/trace_timeline.py $OUT/r05a_prio2_trace_tail.csv $OUT/r05a_prio2_timeline.txt
