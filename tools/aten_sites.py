"""Which Python lines put ATen launches (and hipMemcpy nodes) into the captured stage graphs of the benchmarked training step?
A TorchDispatchMode is switched on inside every `torch.cuda.graph(...)` capture of GraphedPipelined (the torch profiler has no Python
stacks in this image), so what is listed is exactly what the replayed step launches besides the C-ABI kernels: op, operand dtypes /
shapes, innermost frame inside this repo, per stage graph.  Run on the GPU box:

    python tools/aten_sites.py > gpurun_out/aten_sites.txt
"""
import os
import sys
import traceback
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VIEWS = {"view", "reshape", "permute", "expand", "slice", "select", "t", "transpose", "as_strided", "unsqueeze", "squeeze", "detach", "alias",
         "_unsafe_view", "empty", "new_empty", "empty_like", "empty_strided", "split", "unbind", "size", "stride", "lift_fresh", "contiguous",
         "narrow", "view_as", "expand_as", "unfold", "_reshape_alias", "split_with_sizes", "chunk", "numel", "dim", "storage_offset",
         "record_stream", "is_pinned", "_record_function_enter_new", "_record_function_exit"}


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        base = str(func).replace("aten.", "").replace("profiler.", "").split(".")[0]
        if base not in VIEWS and not base.startswith(("sym_", "is_", "_local_scalar")):
            site = "(autograd engine / no frame in omni3d_amd)"
            for fr in reversed(traceback.extract_stack()):
                if "/omni3d_amd/" in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break

            def sh(a):
                return (str(a.dtype).replace("torch.", "") + str(tuple(a.shape))) if isinstance(a, torch.Tensor) else None
            shapes = ",".join(x for x in (sh(a) for a in args[:3]) if x)
            self.sites[(site, str(func).replace("aten.", "") + "  " + shapes[:80])] += 1
        return func(*args, **(kwargs or {}))


def main():
    from omni3d_amd import bench_train as BT
    cfg, model, opt, priors = BT.build(1)
    batch, packed = BT.stage_batch(model, priors, 0)
    recs = []
    enter, exit_ = torch.cuda.graph.__enter__, torch.cuda.graph.__exit__

    def g_enter(self):
        out = enter(self)
        self._rec = Rec()
        self._rec.__enter__()
        return out

    def g_exit(self, *a):
        self._rec.__exit__(None, None, None)
        recs.append(self._rec)
        return exit_(self, *a)
    torch.cuda.graph.__enter__, torch.cuda.graph.__exit__ = g_enter, g_exit
    from omni3d_amd.cubercnn.solver.graphed import GraphedPipelined
    stepper = GraphedPipelined(model, opt, batch, packed, graphs=True)
    torch.cuda.graph.__enter__, torch.cuda.graph.__exit__ = enter, exit_
    # label the captures: M0, W0, M1, W1, ... in capture order (a stage without weight gradients has no W graph)
    labels, it = [], iter(range(len(recs)))
    for k, (gm, gw) in enumerate(stepper.stages):
        labels.append(f"M{k}")
        if gw is not None:
            labels.append(f"W{k}")
    total = 0
    for lab, rec in zip(labels, recs):
        n = sum(rec.sites.values())
        total += n
        print(f"== graph {lab}: {n} ATen ops")
        for (site, op), c in sorted(rec.sites.items(), key=lambda kv: (kv[0][0], kv[0][1])):
            print(f"{c:3d}  {op:110s} {site}")
    print(f"== {total} ATen ops in the {len(recs)} captured graphs of one step")
    # the eager tail of the step (bench_train.run_train.finish)
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    losses, tot, pending = stepper()
    guard = StepGuard(list(losses), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, "cuda")
    opt.skip_flag = guard.skip
    tail = Rec()
    with tail:
        losses, tot, pending = stepper()
        opt.all_reduce_finish(pending, defer_scale=True)
        opt.check_nonfinite(guard.nonfinite_flag)
        guard.update(losses, sync=False)
        opt.step()
    print(f"== eager tail (replay launches + all_reduce_finish + non-finite scan + guard + update): {sum(tail.sites.values())} ATen ops")
    for (site, op), c in sorted(tail.sites.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        print(f"{c:3d}  {op:110s} {site}")


if __name__ == "__main__":
    main()
