"""What does an iteration on a NEVER-SEEN size bucket cost?  The reference's loader draws a new short edge per image
(configs/Base.yaml:10-13), so the drop-in loop keeps meeting new padded shapes: this prints, for the first iterations of the
non-recycled synthetic stream (bench_train.dropin_loop_multiscale_stream's batches), the wall time of every iteration and what it
was -- eager on a new bucket, eager on a bucket seen before, capture, replay -- and the per-kind means.
usage (GPU box): python tools/bench_new_shape.py [iterations] > gpurun_out/new_shape.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    from omni3d_amd import bench_train as BT
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    from omni3d_amd.d2.solver import build_lr_scheduler
    cfg, model, opt, priors = BT.build(1, seed=2)
    sched = build_lr_scheduler(cfg, opt)
    auto = model.__dict__.get("_omni_auto")
    stream = [synthetic.make_multiscale_batch(BT.IMS_PER_GPU, 7000 + s, priors=priors) for s in range(iters)]
    guard = None
    seen = set()
    rows = []
    import gc
    mode = os.environ.get("NEW_SHAPE_GC", "")          # "" | "freeze" | "off": is the bimodal host time the cyclic garbage collector?
    if mode == "freeze":
        gc.collect()
        gc.freeze()
    elif mode == "off":
        gc.disable()
    print("gc:", mode or "default", gc.get_threshold(), "tracked objects:", len(gc.get_objects()), flush=True)
    for it, batch in enumerate(stream):
        sig = auto.signature(batch, record=False) if auto is not None else tuple(max(b["image"].shape[-2 + d] for b in batch) for d in (0, 1))
        cap0, rep0 = (auto.captures, auto.replays) if auto is not None else (0, 0)
        m0 = torch.cuda.memory_reserved()
        torch.cuda.synchronize()
        phases = os.environ.get("NEW_SHAPE_PHASES") == "1"       # synchronize between the phases: where does the time go?
        t0 = time.perf_counter()
        loss_dict = model(batch)
        losses = sum(loss_dict.values())
        t_host_fwd = time.perf_counter()
        if phases:
            torch.cuda.synchronize()
        t_fwd = time.perf_counter()
        if guard is None:
            guard = StepGuard(list(loss_dict), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, losses.device)
            opt.skip_flag = guard.skip
        opt.zero_grad()
        losses.backward()
        t_host_bwd = time.perf_counter()
        if phases:
            torch.cuda.synchronize()
        t_bwd = time.perf_counter()
        opt.all_reduce_grads()
        opt.check_nonfinite(guard.nonfinite_flag)
        guard.update(loss_dict, sync=True)
        opt.step()
        sched.step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        if phases:
            print(f"       fwd host {1e3 * (t_host_fwd - t0):6.1f} + drain {1e3 * (t_fwd - t_host_fwd):6.1f} | bwd host {1e3 * (t_host_bwd - t_fwd):6.1f} + drain "
                  f"{1e3 * (t_bwd - t_host_bwd):6.1f} | rest {1e3 * (time.perf_counter() - t_bwd):6.1f}")
        kind = ("capture" if auto is not None and auto.captures > cap0 else "replay" if auto is not None and auto.replays > rep0
                else "eager_seen" if sig in seen else "eager_new")
        seen.add(sig)
        rows.append((it, kind, sig, ms, (torch.cuda.memory_reserved() - m0) / 2 ** 20))
        print(f"{it:4d} {kind:11s} {str(sig):18s} {ms:8.1f} ms  +{rows[-1][4]:8.0f} MiB reserved", flush=True)
    print()
    for kind in ("eager_new", "eager_seen", "capture", "replay"):
        v = [r[3] for r in rows if r[1] == kind]
        if v:
            v.sort()
            print(f"{kind:11s} n={len(v):3d}  median {v[len(v) // 2]:8.1f} ms  mean {sum(v) / len(v):8.1f} ms  min {v[0]:8.1f}  max {v[-1]:8.1f}")
    if auto is not None:
        print("cache:", auto.stats())
    print("reserved GiB:", round(torch.cuda.memory_reserved() / 2 ** 30, 1), " allocated GiB:", round(torch.cuda.memory_allocated() / 2 ** 30, 1))


if __name__ == "__main__":
    main()
