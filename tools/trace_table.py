#!/usr/bin/env python
"""Per-(kernel, grid) table of a rocprofv3 --kernel-trace CSV: launches per step, average / total duration, share.
usage: trace_table.py <kernel_trace.csv> <executions_of_the_step> [top]"""
import csv
import sys
from collections import defaultdict

path, execs = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 70
acc = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name")
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y") or "", r.get("Grid_Size_Z") or "")
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = acc[(name, g)]
        a[0] += 1
        a[1] += dur
tot = sum(v[1] for v in acc.values())
print(f"total kernel time {tot / execs / 1e6:.3f} ms / step over {sum(v[0] for v in acc.values()) / execs:.0f} launches / step")
print(f"{'kernel':62s} {'grid':>22s} {'n/step':>7s} {'avg us':>8s} {'us/step':>9s} {'%':>5s}")
for (name, g), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{name:62s} {'x'.join(x for x in g if x):>22s} {n / execs:7.2f} {t / n / 1e3:8.1f} {t / execs / 1e3:9.1f} {100 * t / tot:5.2f}")
