#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/r03q_bench.log 2> $OUT/r03q_bench.err; echo "bench rc=$?"; tail -3 $OUT/r03q_bench.err | cut -c1-300
python - <<'PY'
import json
for ln in open("gpurun_out/r03q_bench.log"):
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]
        print("images/s", d["value"], "ms", d["ms_per_step"], "dropin", d["dropin_loop_ms_per_step"])
        print("roofline:", r["kernel"][:200]); print("  frac", r["frac"], "achieved", r["achieved"], "traffic", r["traffic"], "alg bytes", r["algorithmic_bytes_per_launch"], "mfma busy", r["pmc_mfma_busy_frac"])
        for f in r["families"]: print("  %-70s %7.3f ms %6.1f TF %.2f busy %s share %s traffic %s"%(f["family"][:70],f["kernel_ms"],f["tflops"],f["frac"],f.get("pmc_mfma_busy_frac") and round(f["pmc_mfma_busy_frac"],2),f.get("share_of_kernel_time_in_table") and round(f["share_of_kernel_time_in_table"],3), f.get("pmc_traffic_bytes")))
        io=d["iou3d"]; print("iou3d", io["value"], io["roofline"]["kernel_ms"], io["roofline"]["traffic"], io["roofline"]["valu"]["valu_active_frac_of_wave_cycles"], io["roofline"]["valu"]["source"])
PY
