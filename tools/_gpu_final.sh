#!/bin/bash
# round 5, last call: the GPU test suite at the final head, then the measurement set (tools/profile_final.sh)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05z_gpu_tests.log; cat gpurun_out/r05z_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r05z_smoke.log
bash tools/profile_final.sh r05z
