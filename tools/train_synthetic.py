#!/usr/bin/env python
"""The step loop of the reference's tools/train_net.py:do_train (:176-313) on the HIP path, written against the REFERENCE's
import paths (omni3d_amd.install()) and fed with synthetic Omni3D-shaped batches instead of the data pipeline (out of scope,
SURVEY.md 8b): model(data) -> sum(losses).backward() -> non-finite check -> optimizer.step() -> scheduler.step(), EventStorage
scalars, PeriodicCheckpointerOnlyOne.  Nothing in this loop knows about graphs: after two iterations with the same batch shape
`model(data)` switches to the staged hipGraph replay by itself (cubercnn/solver/autoreplay.py; OMNI_AUTO_REPLAY=0 keeps eager
launches).

    python tools/train_synthetic.py --iters 60 [--config cubercnn_ResNet34_FPN.yaml] [--out /tmp/run]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import omni3d_amd

omni3d_amd.install()
from cubercnn.config import get_cfg_defaults                       # noqa: E402  (reference import paths from here on)
from cubercnn.modeling.backbone import build_dla_from_vision_fpn_backbone   # noqa: E402,F401  (registrations, as in
from cubercnn.modeling.meta_arch import build_model                 # noqa: E402        tools/train_net.py:43-47)
from cubercnn.modeling.proposal_generator import RPNWithIgnore      # noqa: E402,F401
from cubercnn.modeling.roi_heads import ROIHeads3D                  # noqa: E402,F401
from cubercnn.solver import PeriodicCheckpointerOnlyOne, build_optimizer   # noqa: E402
from detectron2.config import get_cfg                               # noqa: E402
from detectron2.solver import build_lr_scheduler                    # noqa: E402
from detectron2.utils.events import EventStorage                    # noqa: E402


class _Checkpointer:
    def __init__(self, model, optimizer, scheduler, out):
        self.model, self.optimizer, self.scheduler, self.out = model, optimizer, scheduler, out

    def save(self, name, **extra):
        if self.out:
            os.makedirs(self.out, exist_ok=True)
            torch.save({"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
                        "scheduler": self.scheduler.state_dict(), **extra}, os.path.join(self.out, name + ".pth"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--config", default="cubercnn_DLA34_FPN.yaml")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from omni3d_amd import synthetic
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", args.config))
    cfg.merge_from_list(["MODEL.WEIGHTS", "synthetic://random-init", "VIS_PERIOD", 0, "SOLVER.IMS_PER_BATCH", args.batch,
                         "SOLVER.BASE_LR", 0.12 * args.batch / 192.0, "SOLVER.MAX_ITER", args.iters,
                         "SOLVER.STEPS", (int(0.6 * args.iters), int(0.8 * args.iters)), "SOLVER.WARMUP_ITERS", max(args.iters // 10, 1),
                         "SOLVER.CHECKPOINT_PERIOD", max(args.iters // 2, 1), "MODEL.DEVICE", os.environ.get("OMNI_DEVICE", "cuda")])
    priors = synthetic.make_priors(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
    torch.manual_seed(0)
    model = build_model(cfg, priors)
    model.train()
    optimizer = build_optimizer(cfg, model)
    scheduler = build_lr_scheduler(cfg, optimizer)
    ckpt = PeriodicCheckpointerOnlyOne(_Checkpointer(model, optimizer, scheduler, args.out), cfg.SOLVER.CHECKPOINT_PERIOD,
                                       max_iter=cfg.SOLVER.MAX_ITER)
    if cfg.MODEL.DEVICE == "cpu":
        print("built on the CPU (dry run: the kernels need the GPU)")
        return
    from cubercnn.solver import StepGuard
    guard = None
    pool = [synthetic.make_batch(args.batch, args.size, args.size, num_gt=8, seed=s, priors=priors) for s in range(4)]
    t0 = time.perf_counter()
    with EventStorage(0) as storage:
        for it in range(cfg.SOLVER.MAX_ITER):
            storage.iter = it
            data = pool[it % len(pool)]
            loss_dict = model(data)
            losses = sum(loss_dict.values())
            if guard is None:                            # :157-170 state; the skip flag is read by the fused SGD kernel
                guard = StepGuard(list(loss_dict), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, losses.device)
                optimizer.skip_flag = guard.skip
            optimizer.zero_grad()
            losses.backward()
            optimizer.all_reduce_grads()
            optimizer.check_nonfinite(guard.nonfinite_flag)                       # :222-233 as one pass
            skipped, retry, reduced = guard.update(loss_dict)                       # :186-215, :237-270: ONE small all-reduce
            storage.put_scalars(**{k.replace("/", "_"): v for k, v in reduced.items()})
            optimizer.step()                                                        # skipped on the device when diverging
            if not skipped:
                storage.put_scalar("lr", optimizer.param_groups[0]["lr"], smoothing_hint=False)
            if retry:
                print(f"!! restart requested at iteration {it} (exploding loss) !!")   # :272-285 returns False here
                return False
            scheduler.step()
            if not skipped:
                ckpt.step(it)
            if it % 10 == 0 or it == cfg.SOLVER.MAX_ITER - 1:
                print(f"iter {it:4d}  total {reduced['total_loss']:8.4f}  lr {optimizer.param_groups[0]['lr']:.6f}  skipped {int(skipped)}  "
                      + "  ".join(f"{k.split('/')[-1]} {v:.3f}" for k, v in reduced.items() if k != "total_loss"), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    auto = getattr(model, "_omni_auto", None)
    mode = (f"{auto.replays} of them replayed as staged hipGraphs from inside model(data)" if auto is not None and auto.replays
            else "eager launches" + (f" ({auto.failed})" if auto is not None and auto.failed else ""))
    print(f"done: {cfg.SOLVER.MAX_ITER} iterations, {cfg.SOLVER.MAX_ITER * args.batch / dt:.1f} images/s including the capture, host-side "
          f"batch packing and logging; {mode}")


if __name__ == "__main__":
    main()
