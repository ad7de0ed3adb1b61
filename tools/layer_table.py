"""Per-launch table of one eager training step: every C-ABI call is bracketed by HIP events; calls are grouped by (entry point,
problem size) and listed with their time, executed TFLOP/s (MFMA entry points) and share of the step.

    python tools/layer_table.py > gpurun_out/layer_table.txt
"""
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from omni3d_amd import bench_train as BT
    from omni3d_amd import lib
    from omni3d_amd.functional import total_loss
    from omni3d_amd.profile_io import ExecutedFlops
    cfg, model, opt, priors = BT.build(1)
    batch, packed = BT.stage_batch(model, priors, 0)

    def step():
        opt.zero_grad()
        losses = model(batch, packed)
        total_loss(losses).backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L = lib.get()
    orig = L.call
    recs = []

    def call(name, *a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(name, *a)
        e.record()
        ints = tuple(x for x in a if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 24))
        recs.append((name, ints, ExecutedFlops._flops(name, a), s, e))
        return r
    L.call = call
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    step()
    t1.record()
    torch.cuda.synchronize()
    L.call = orig
    groups = OrderedDict()
    tot = 0.0
    for name, ints, fl, s, e in recs:
        us = s.elapsed_time(e) * 1e3
        tot += us
        g = groups.setdefault((name, ints), [0, 0.0, 0.0])
        g[0] += 1
        g[1] += us
        g[2] += fl
    print(f"eager step {t0.elapsed_time(t1):.2f} ms (host-bound), sum of bracketed C-ABI calls {tot/1e3:.3f} ms, {len(recs)} calls")
    mf = sum(g[1] for (n, _), g in groups.items() if g[2] > 0)
    print(f"MFMA entry points: {mf/1e3:.3f} ms, {sum(g[2] for g in groups.values())/1e9:.1f} GFLOP executed")
    print(f"{'entry':28s} {'n':>3s} {'us/call':>9s} {'us tot':>9s} {'TF/s':>7s}  dims")
    for (name, ints), (n, us, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:120]:
        tf = f"{fl / us / 1e6:7.1f}" if fl > 0 else "       "
        print(f"{name[5:33]:28s} {n:3d} {us/n:9.1f} {us:9.1f} {tf}  {ints[:14]}")


if __name__ == "__main__":
    main()
