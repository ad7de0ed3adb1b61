#!/usr/bin/env python
"""Timeline of ONE replayed training step from a rocprofv3 kernel trace tail (tools/profile_step.sh): every kernel of the main queue
and of the weight-gradient queue with its start offset and duration, plus per-phase sums.  usage: trace_timeline.py <tail.csv> [out]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
a, b = sg[-3] + 1, sg[-2]
step = rows[a:b + 1]
t0 = step[0]["s"]
qs = sorted({r["Queue_Id"] for r in step}, key=lambda q: -sum(1 for r in step if r["Queue_Id"] == q))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(f"# one step: {len(step)} kernels, wall {(step[-1]['e'] - t0) / 1e6:.3f} ms; queues {qs}", file=out)
for q in qs:
    rr = [r for r in step if r["Queue_Id"] == q]
    print(f"## queue {q}: {len(rr)} kernels, busy {sum(r['e'] - r['s'] for r in rr) / 1e6:.3f} ms, span {(rr[0]['s'] - t0) / 1e6:.3f} .. {(rr[-1]['e'] - t0) / 1e6:.3f} ms", file=out)
    prev = None
    for r in rr:
        gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
        g = "x".join(str(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if r.get(k) not in (None, "", "1") or k == "Grid_Size_X")
        print(f"{(r['s'] - t0) / 1e3:9.1f} us  +{(r['e'] - r['s']) / 1e3:7.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:56]:56s} {g}", file=out)
        prev = r["e"]
