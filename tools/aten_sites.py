"""Which Python lines launch the small ATen kernels (fill / add / mul / copy / cat) of one eager training step?
torch.profiler with Python stacks, grouped by the innermost frame inside this repo.  Run on the GPU box:

    python tools/aten_sites.py > gpurun_out/aten_sites.txt
"""
import argparse
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from omni3d_amd import bench_train as BT
    cfg, model, opt, priors = BT.build(1)
    batch, packed = BT.stage_batch(model, priors, 0)

    def step():
        opt.zero_grad()
        losses = model(batch, packed)
        total = sum(losses.values())
        total.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    want = ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::cat", "aten::clone", "aten::zeros",
            "aten::div", "aten::sum", "aten::contiguous", "aten::neg", "aten::where", "aten::clamp", "aten::clamp_", "aten::index", "aten::stack")
    sites = defaultdict(int)
    for ev in prof.events():
        if ev.name not in want:
            continue
        stack = ev.stack or []
        frame = next((f for f in stack if "/omni3d_amd/" in f or "bench" in f or "tools/" in f), stack[0] if stack else "(no python stack: autograd engine)")
        sites[(ev.name, frame.replace(ROOT + "/", "")[:120], "")] += 1
    tot = defaultdict(int)
    for (name, _, _), n in sites.items():
        tot[name] += n
    print("kernel launches per op:", dict(sorted(tot.items(), key=lambda kv: -kv[1])))
    for (name, frame, back), n in sorted(sites.items(), key=lambda kv: -kv[1])[:90]:
        if n:
            print(f"{n:4d}  {name:16s} {back:9s} {frame}")


if __name__ == "__main__":
    main()
