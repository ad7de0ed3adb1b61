"""Which Python lines issue the small ATen ops of one training step?  Runs the step on the CPU under the host emulator with a
TorchDispatchMode that records every aten op that would be a device launch, together with the innermost frame inside this repo (the
torch profiler has no Python stacks in this image).  usage: python tools/aten_sites_cpu.py"""
import os
import sys
import traceback
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from omni3d_amd import lib as L

L._install_for_tests(L.HipLibrary(os.path.join(ROOT, "tests", "hipemu", "libomni3d_emu.so"), emulated=True))
from oracle import make_golden as MG
from omni3d_amd import synthetic
from omni3d_amd.cubercnn.solver import build_optimizer

VIEWS = ("view", "reshape", "permute", "expand", "slice", "select", "t.default", "transpose", "as_strided", "unsqueeze", "squeeze", "detach",
         "alias", "_unsafe_view", "empty", "new_empty", "split", "unbind", "is_", "size", "stride", "sym_", "_local_scalar", "lift_fresh", "contiguous")


VIEWS_EXACT = {"view", "reshape", "permute", "expand", "slice", "select", "t", "transpose", "as_strided", "unsqueeze", "squeeze", "detach", "alias",
               "_unsafe_view", "empty", "new_empty", "empty_like", "empty_strided", "split", "unbind", "size", "stride", "lift_fresh", "contiguous",
               "narrow", "view_as", "expand_as", "unfold", "_reshape_alias", "split_with_sizes", "chunk", "numel", "dim", "storage_offset"}


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.replace("aten.", "").split(".")[0]
        if base not in VIEWS_EXACT and not base.startswith(("sym_", "is_", "_local_scalar")):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/omni3d_amd/" in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            def sh(a):
                return (str(a.dtype).replace("torch.", "") + str(tuple(a.shape))) if isinstance(a, torch.Tensor) else None
            shapes = ",".join(x for x in (sh(a) for a in args[:3]) if x)
            self.sites[(site, name.replace("aten.", "") + "  " + shapes[:70])] += 1
        return func(*args, **(kwargs or {}))


LIGHT = ["MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 100,
         "MODEL.RPN.POST_NMS_TOPK_TRAIN", 30, "MODEL.DLA.TYPE", "dla46_c", "MODEL.FPN.OUT_CHANNELS", 32, "MODEL.ROI_BOX_HEAD.FC_DIM", 64,
         "MODEL.ROI_CUBE_HEAD.FC_DIM", 64]
priors = synthetic.make_priors(50)
cfg = MG.product_cfg(LIGHT)
model = MG.build_product_model(cfg, priors, 11, device="cpu")
model.train()
opt = build_optimizer(cfg, model)
batch = synthetic.make_batch(int(os.environ.get("ATEN_SITES_B", "2")), 64, 64, num_gt=3, seed=40, priors=priors)
packed = model.prepack(batch)
from omni3d_amd.d2.events import EventStorage
from omni3d_amd.cubercnn.solver.graphed import GraphedPipelined
from omni3d_amd.cubercnn.solver.guard import StepGuard
from omni3d_amd.functional import total_loss

# the step bench.py times (omni3d_amd/bench_train.py): staged stepper (eager launches here) + non-finite scan + guard + fused update
with EventStorage(0):
    stepper = GraphedPipelined(model, opt, batch, packed, graphs=False)
    names = None
    guard = None

    def finish(losses, total):
        global guard
        if guard is None:
            guard = StepGuard(list(losses), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, "cpu")
            opt.skip_flag = guard.skip
        opt.check_nonfinite(guard.nonfinite_flag)
        guard.update(losses, sync=False)
        opt.step()

    for _ in range(2):
        losses, total, pending = stepper()
        opt.all_reduce_finish(pending, defer_scale=True)
        finish(losses, total)
    a, b = Rec(), Rec()
    with a:
        losses, total, pending = stepper()
    with b:
        opt.all_reduce_finish(pending, defer_scale=True)
        finish(losses, total)
for title, rec in (("staged forward + backward (what the seven stage graphs capture)", a), ("all_reduce_finish + non-finite scan + guard + update", b)):
    print(f"== {title}: {sum(rec.sites.values())} non-view aten ops")
    for (site, op), n in sorted(rec.sites.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        print(f"{n:3d}  {op:100s} {site}")
