#!/bin/bash
# A/B of environment knobs on one GPU box: every variant twice, interleaved, step time + M0 end per run.
# usage: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one variant's environment; "-" = default)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  r=$(env $e OMNI_BENCH_CONDITION_STEPS=50 OMNI_BENCH_SKIP_STAGE_ENDS=1 OMNI_PIPE_TIMING=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 300 python bench.py --workload train --steps 30 --warmup 5 2>&1 | grep -E "pipe timing|images/sec")
  echo "[$v] $(echo "$r" | grep -o 'M0 end [0-9.]* ms') $(echo "$r" | grep -o '"ms_per_step": [0-9.]*')"
done
done
