#!/usr/bin/env python
"""Condenses rocprofv3 --pmc counter_collection CSVs (one pass per counter) into a small per-kernel table.
usage: pmc_summary.py <out.csv> <counter_collection.csv> [...]   (only the omni3d conv / GEMM kernels are kept)"""
import csv
import re
import sys
from collections import defaultdict

out, files = sys.argv[1], sys.argv[2:]
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(conv_(?:fwd|dgrad|wgrad)_kernel<[^>]*>|gemm_nt_persistent_kernel)", n)
        if m:
            acc[(m.group(1) + " grid=" + r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(out, "w") as fh:
    fh.write("kernel,counter,launches,mean_value_KB,mean_MB\n")
    for (k, c), v in sorted(acc.items()):
        mean = sum(v) / len(v)
        fh.write(f"\"{k}\",{c},{len(v)},{mean:.1f},{mean * 1024 / 1e6:.1f}\n")
print(open(out).read())
