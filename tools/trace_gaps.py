"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: per training step (one period between consecutive roi_sample_kernel launches), the wall
time, the summed kernel time, the idle time between consecutive kernels and which kernels the idle time sits in front of.

    python tools/trace_gaps.py <..._kernel_trace.csv> [out.txt]
"""
import csv
import sys
from collections import defaultdict


def main(path, out=None):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "roi_sample_kernel" in r[2]]      # exactly one launch per training step
    lines = []
    if len(ends) < 4:
        print("not enough steps")
        return
    # the last 8 steps (graph replays)
    sel = ends[-9:]
    walls, busy, gaps, nk = [], [], [], []
    gap_by = defaultdict(lambda: [0, 0.0])
    dur_by = defaultdict(lambda: [0, 0.0])
    overlap = 0.0
    for a, b in zip(sel[:-1], sel[1:]):
        seg = rows[a + 1: b + 1]
        walls.append((seg[-1][1] - seg[0][0]) / 1e6)
        busy.append(sum(e - s for s, e, _ in seg) / 1e6)
        nk.append(len(seg))
        g = 0.0
        cur_end = seg[0][1]
        for (s, e, n) in seg[1:]:
            d = s - cur_end
            if d > 0:
                g += d
                gap_by[n.split("(")[0][-60:]][0] += 1
                gap_by[n.split("(")[0][-60:]][1] += d
            else:
                overlap += -d if e > cur_end else (e - s)
            cur_end = max(cur_end, e)
        for (s, e, n) in seg:
            k = n.split("(")[0][-60:]
            dur_by[k][0] += 1
            dur_by[k][1] += e - s
        gaps.append(g / 1e6)
    S = len(walls)
    lines.append(f"steps analysed {S}: wall ms/step {sum(walls)/S:.3f}  kernel-busy ms/step {sum(busy)/S:.3f}  idle ms/step {sum(gaps)/S:.3f}  "
                 f"kernels/step {sum(nk)/S:.0f}  overlapped ms/step {overlap/1e6/S:.3f}")
    lines.append("idle time in front of (top 25): name, count/step, idle us/step, mean gap us")
    for k, (c, d) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f"  {k:60s} {c/S:7.1f} {d/1e3/S:9.1f} {d/1e3/max(c,1):7.2f}")
    lines.append("kernel time (top 25): name, count/step, us/step, mean us")
    for k, (c, d) in sorted(dur_by.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f"  {k:60s} {c/S:7.1f} {d/1e3/S:9.1f} {d/1e3/max(c,1):7.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
