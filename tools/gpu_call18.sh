#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
OMNI_SLOW=1 timeout 1200 python -m pytest tests/test_zy_backbone_variants.py tests/test_zz_backbone_model_parity.py -m gpu -q --durations=5 > $OUT/r03r_slow_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/r03r_slow_gpu.log | cut -c1-220
OMNI_BENCH_SKIP_CPU=1 timeout 300 python bench.py --workload infer 2>/dev/null | cut -c1-700 | tee $OUT/r03r_infer.log
OMNI_BENCH_CONFIG=cubercnn_ResNet34_FPN.yaml OMNI_BENCH_SKIP_CPU=1 OMNI_BENCH_SKIP_ROOFLINE=1 timeout 300 python bench.py --workload train 2>/dev/null | python -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('ResNet34: %.2f images/s  %.3f ms' % (d['value'], d['ms_per_step']))" | tee $OUT/r03r_resnet34.log
