#!/usr/bin/env python
"""Runs `rocprofv3 --pmc` passes (one pass per counter group, no tracing besides --kernel-trace) around a command and
condenses the per-dispatch rows into one table per (kernel, grid).

usage: pmc_run.py <outdir> <summary.csv> [--filter REGEX] -- <command ...>

Counter groups follow MI355X_MICROARCH.md "rocprofv3 PMC slots" (SQ 8 slots, FETCH_SIZE and WRITE_SIZE in separate passes);
names that `rocprofv3 -L` does not list on this box are dropped from their group instead of failing the pass.
FETCH_SIZE is reported raw AND doubled (gfx950 counts wide coalesced reads at 1/2, see the guide's HBM section).
Derived columns (cycle based, so DVFS independent):
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * n_SIMD_units_reporting)   [see note in profiles/README.md]
  mfma_flop        = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512   (MOPS counts units of 512 flops)
"""
import csv
import os
import re
import subprocess
import sys
from collections import defaultdict

GROUPS = [
    ["SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
     "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
     "SQ_INSTS_MFMA", "SQ_WAVES"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    # lane utilisation of the VALU: thread-cycles the VALU spent on ACTIVE lanes against the instruction count
    ["SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
]


def available():
    try:
        txt = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120).stdout
    except Exception as e:  # noqa: BLE001
        print("rocprofv3 -L failed:", e)
        return None
    return set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", txt))


def main():
    argv = sys.argv[1:]
    sep = argv.index("--")
    head, cmd = argv[:sep], argv[sep + 1:]
    outdir, summary = head[0], head[1]
    filt = re.compile(head[head.index("--filter") + 1]) if "--filter" in head else re.compile(r"omni|conv_|gemm_|wino|bn_|stem_|iou_|head16|dgrad_|roi_align")
    names = available()
    os.makedirs(outdir, exist_ok=True)
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for gi, group in enumerate(GROUPS):
        g = [c for c in group if names is None or c in names]
        dropped = [c for c in group if c not in g]
        if dropped:
            print(f"pass {gi}: counters not listed by rocprofv3 -L, dropped: {dropped}")
        if not g:
            continue
        d = os.path.join(outdir, f"pass{gi}")
        full = ["rocprofv3", "--kernel-trace", "--pmc", *g, "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd
        print("running:", " ".join(full), flush=True)
        r = subprocess.run(full, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {gi} failed rc={r.returncode}: {r.stderr[-800:]}")
            continue
        for root, _, files in os.walk(d):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    for row in csv.DictReader(open(os.path.join(root, f))):
                        n = row["Kernel_Name"]
                        if not filt.search(n):
                            continue
                        short = re.sub(r"\(anonymous namespace\)::", "", n)
                        short = re.sub(r"\(.*$", "", short).replace("void ", "")
                        key = (short, row["Grid_Size"])
                        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"],
                                     row["Scratch_Size"])
    counters = sorted({c for v in acc.values() for c in v})
    with open(summary, "w") as fh:
        fh.write("kernel,grid,launches,vgpr,agpr,sgpr,lds,scratch," + ",".join(counters) + ",FETCH_SIZE_x2_MB,WRITE_SIZE_MB,mfma_flop\n")
        for key in sorted(acc):
            v = acc[key]
            mean = {c: (sum(v[c]) / len(v[c]) if v.get(c) else float("nan")) for c in counters}
            n = max(len(x) for x in v.values())
            f2 = mean.get("FETCH_SIZE", float("nan")) * 2 * 1024 / 1e6
            wr = mean.get("WRITE_SIZE", float("nan")) * 1024 / 1e6
            fl = mean.get("SQ_INSTS_VALU_MFMA_MOPS_F32", float("nan")) * 512
            fh.write(f"\"{key[0]}\",{key[1]},{n}," + ",".join(meta[key]) + "," + ",".join(f"{mean[c]:.1f}" for c in counters) +
                     f",{f2:.1f},{wr:.1f},{fl:.4g}\n")
    print(open(summary).read())


if __name__ == "__main__":
    main()
