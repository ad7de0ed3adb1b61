#!/bin/bash
# A/B of two source trees on ONE GPU box (step time differs by +-1 % between boxes, so before / after must share a box):
# usage: tools/ab_trees.sh <other_tree> [reps]   -- alternates `bench.py --workload train` in this tree and in <other_tree>
# (a built copy of another commit: `git archive <commit> | tar -x -C _old && make -C _old/omni3d_amd/csrc`; _old/ is git-ignored)
set -u
OTHER=${1:-_old}; REPS=${2:-2}; HERE=$(pwd)
run() { (cd $1 && OMNI_BENCH_CONDITION_STEPS=50 OMNI_BENCH_SKIP_STAGE_ENDS=1 OMNI_PIPE_TIMING=1 OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 \
        timeout 300 python bench.py --workload train --steps 30 --warmup 5 2>&1 | grep -E "pipe timing|images/sec" \
        | python -c "
import sys, re, json
t = sys.stdin.read()
m0 = re.search(r'M0 end ([0-9.]+)', t); ends = re.findall(r'[MW]6? end ([0-9.]+) ms', t)
j = [l for l in t.splitlines() if l.startswith('{')]
r = json.loads(j[-1]) if j else {}
print('  M0 end', m0.group(1) if m0 else '?', ' windows', r.get('windows', {}).get('ms_per_step'), ' median', r.get('ms_per_step'))
"); }
for rep in $(seq $REPS); do
  echo "[this tree]"; run $HERE
  echo "[$OTHER]"; run $HERE/$OTHER
done
