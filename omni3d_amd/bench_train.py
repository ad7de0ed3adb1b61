"""Training-throughput workload of bench.py: cubercnn_DLA34_FPN, batch 4 per GPU, synthetic
512x512 Omni3D-shaped inputs pre-staged in HBM (BASELINE.json configs[1]; configs[2] when launched
on N GPUs).  One step = preprocess + forward + 10 losses + backward + (RCCL all-reduce of the flat
gradient bucket) + fused non-finite scan + fused SGD-momentum update, fp32 throughout."""
import os
import time

import torch
import torch.distributed as dist

from .functional import total_loss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md: f32-input MFMA, dense
TRAIN_GFLOP_PER_IMAGE = 307.8      # fwd 103.0 GFLOP, fwd+dgrad+wgrad 307.8 GFLOP
IMS_PER_GPU = int(os.environ.get("OMNI_BENCH_IMS", "4"))        # BASELINE: 4 images / GPU
IMAGE_SIZE = int(os.environ.get("OMNI_BENCH_SIZE", "512"))       # BASELINE: 512 x 512
DEVICE = os.environ.get("OMNI_BENCH_DEVICE", "cuda")             # "cpu" only in the GPU-less CI (tests/test_bench_cli.py), see bench.py
# extra `KEY value ...` config overrides; like OMNI_BENCH_IMS / _SIZE they make the line a functional check (`nonstandard`)
OVERRIDES = os.environ.get("OMNI_BENCH_OVERRIDES", "").split()
NONSTANDARD = IMS_PER_GPU != 4 or IMAGE_SIZE != 512 or bool(OVERRIDES) or DEVICE != "cuda"


def _sync():
    if DEVICE == "cuda":
        torch.cuda.synchronize()


# BASELINE.json configs[1] by default; OMNI_BENCH_CONFIG=cubercnn_ResNet34_FPN.yaml selects configs[3]'s model
CONFIG = os.environ.get("OMNI_BENCH_CONFIG", "cubercnn_DLA34_FPN.yaml")
MODEL_NAME = CONFIG.replace("cubercnn_", "").replace(".yaml", "")


def build(world, device=DEVICE, seed=0):
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.config import get_cfg_defaults
    from omni3d_amd.cubercnn.modeling import backbone, proposal_generator, roi_heads  # noqa: F401 (registrations)
    from omni3d_amd.cubercnn.modeling.meta_arch import build_model
    from omni3d_amd.cubercnn.solver import build_optimizer
    from omni3d_amd.d2.config import get_cfg
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", CONFIG))
    ims = IMS_PER_GPU * world
    # README.md:123-132 scaling rule of the reference: lr scales with the batch (0.12 at 192 images)
    cfg.merge_from_list(["MODEL.DEVICE", device, "VIS_PERIOD", 0, "MODEL.WEIGHTS", "synthetic://random-init",
                         "SOLVER.IMS_PER_BATCH", ims, "SOLVER.BASE_LR", 0.12 * ims / 192.0])
    if OVERRIDES:
        cfg.merge_from_list(OVERRIDES)
    priors = synthetic.make_priors(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
    torch.manual_seed(seed)
    model = build_model(cfg, priors)
    model.train()
    opt = build_optimizer(cfg, model)
    return cfg, model, opt, priors


def stage_batch(model, priors, rank, n=IMS_PER_GPU, size=IMAGE_SIZE):
    from omni3d_amd import synthetic
    batch = synthetic.make_batch(n, size, size, num_gt=8, seed=1000 + rank, priors=priors)
    packed = model.prepack(batch)
    for b in batch:
        b["image"] = b["image"].to(model.device)
    return batch, packed


def run_train(args, world, rank):
    main_prio = int(os.environ.get("OMNI_MAIN_PRIORITY", "0"))
    if main_prio != 0 and DEVICE == "cuda":       # A/B knob: the critical path on a high-priority stream (torch: -1 = high)
        torch.cuda.set_stream(torch.cuda.Stream(priority=main_prio))
    cfg, model, opt, priors = build(world)
    if world > 1:
        dist.broadcast(opt.flat_param, src=0)
    batch, packed = stage_batch(model, priors, rank)
    # the loop's safety logic (tools/train_net.py:157-285): rolling-loss divergence test, NaN/Inf gradient scan, skip / retry
    # decisions -- device-side state + ONE 12-float all-reduce per step (cubercnn/solver/guard.py); the fused SGD kernel
    # reads the skip flag on the device, the host only looks at it after the timed region
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    LOSS_NAMES = ["BoxHead/loss_cls", "BoxHead/loss_box_reg", "Cube/uncert", "Cube/loss_dims", "Cube/loss_xy", "Cube/loss_z",
                  "Cube/loss_pose", "Cube/loss_joint", "rpn/cls", "rpn/loc"]
    guard = StepGuard(LOSS_NAMES, cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, DEVICE)
    opt.skip_flag = guard.skip

    def finish(losses, total):
        # (the step's own record lives in the guard: guard.out = [skipped, retry, total loss, 10 losses] of the last step, guard.state[2]
        # = steps skipped so far -- read once before / after the timed windows instead of two copy launches per step)
        opt.check_nonfinite(guard.nonfinite_flag)
        guard.update(losses, sync=False)
        opt.step()

    def eager_step():
        if getattr(model, "feature_cut", None) is not None and hasattr(graphed, "_eager"):    # staged backward installed
            losses, total, pending = graphed._eager()
            opt.all_reduce_finish(pending + graphed._late(graphed._eager_exchanged_all), defer_scale=True)
            return finish(losses, total)
        opt.zero_grad()
        losses = model(batch, packed)
        total = total_loss(losses)             # == sum(losses.values()), two launches
        total.backward()
        opt.all_reduce_finish(opt.all_reduce_begin("early") + opt.all_reduce_begin("late"), defer_scale=True)
        finish(losses, total.detach())

    # zero_grad + forward + losses + backward replayed as one hipGraph (OMNI_BENCH_GRAPH=0: eager launches);
    # all-reduce / non-finite scan / SGD update stay eager (host-side learning rate).  With more than one rank the
    # backward is cut at the FPN features into two graphs so the all-reduce of the heads' gradients (61 % of the
    # bytes) runs beside the backbone's backward (GraphedTwoPhase); the collective sequence is the same on every
    # rank whether or not its capture succeeded.
    use_graph = os.environ.get("OMNI_BENCH_GRAPH", "1") != "0" and DEVICE == "cuda"
    two_phase = world > 1 or os.environ.get("OMNI_BENCH_TWO_PHASE") == "1"
    graphed, graph_note = None, "eager (OMNI_BENCH_GRAPH=0)"
    from omni3d_amd.cubercnn.solver.graphed import GraphedForwardBackward, GraphedPipelined, GraphedTwoPhase
    pipelined = os.environ.get("OMNI_BENCH_PIPELINE", "1") != "0" and os.environ.get("OMNI_BENCH_TWO_PHASE") != "1"
    if world > 1 and DEVICE == "cuda":
        from omni3d_amd.cubercnn.solver import graphed as _graphed_mod
        _graphed_mod.set_pipe_timing(True)       # per-stage device timestamps + exposed exchange time in every N > 1 line
        opt.exchange_timing = True
    if pipelined:
        try:
            graphed = GraphedPipelined(model, opt, batch, packed, graphs=use_graph)
            graph_note = (f"{len(graphed.stages)} backward stages x (critical-path hipGraph on the main stream | weight-gradient hipGraph "
                          "on a second stream), all-reduce of every backward stage's gradient range behind that stage's weight-gradient graph" if use_graph
                          else "eager staged backward, weight gradients on a second stream")
        except Exception as e:   # capture refused: same sequence with eager launches
            graphed = GraphedPipelined(model, opt, batch, packed, graphs=False)
            graph_note = f"eager staged backward (capture failed: {type(e).__name__}: {str(e)[:160]})"
        two_phase = True          # same call protocol: (losses, total, pending)
    elif two_phase:
        try:
            graphed = GraphedTwoPhase(model, opt, batch, packed, graphs=use_graph)
            graph_note = ("two hipGraphs (fwd+heads bwd | backbone bwd), all-reduce of the heads' gradients overlapped"
                          if use_graph else "eager two-phase backward, all-reduce of the heads' gradients overlapped")
        except Exception as e:   # capture refused: same sequence with eager launches
            graphed = GraphedTwoPhase(model, opt, batch, packed, graphs=False)
            graph_note = f"eager two-phase (capture failed: {type(e).__name__}: {str(e)[:160]})"
    elif use_graph:
        try:
            graphed = GraphedForwardBackward(model, opt, batch, packed)
            graph_note = "hipGraph replay of zero_grad+fwd+losses+bwd"
        except Exception as e:   # capture refused: report it, measure the eager path
            graphed, graph_note = None, f"eager (capture failed: {type(e).__name__}: {str(e)[:200]})"

    def graph_step():
        if two_phase:
            losses, total, pending = graphed()
            opt.all_reduce_finish(pending, defer_scale=True)       # 1/world is folded into the SGD kernel
        else:
            losses, total = graphed()
            opt.all_reduce_finish(opt.all_reduce_begin("early") + opt.all_reduce_begin("late"), defer_scale=True)
        finish(losses, total)

    step = graph_step if graphed is not None else eager_step

    # executed (Winograd-aware) flops of one step: count the multiply-adds of every MFMA launch during one eager step
    from omni3d_amd.profile_io import ExecutedFlops
    executed_flops = 0.0
    if DEVICE == "cuda":
        with ExecutedFlops() as counter:
            eager_step()
        executed_flops = counter.flops

    # Conditioning: a fixed number of UNTIMED steps before the warm-up so that every timed window sees the device in
    # the power / clock state of sustained training, not the first 0.3 s after idle (fixed count: every rank issues the same collectives)
    n_cond = int(os.environ.get("OMNI_BENCH_CONDITION_STEPS", "150" if DEVICE == "cuda" else "0"))
    from omni3d_amd.profile_io import GpuState
    gpu = GpuState(torch.cuda.current_device()) if DEVICE == "cuda" else None
    state = {"idle": gpu.sample() if gpu else None}
    for i in range(n_cond):
        step()
        if i % 16 == 15:
            _sync()                              # (keeps the host at most 16 steps ahead)
    for _ in range(args.warmup):
        step()
    # WINDOWS back-to-back windows of exactly `--steps` steps, each bracketed by synchronize + barrier on both sides and reduced with
    # MAX over the ranks; the line's `ms_per_step` / `value` are the MEDIAN window (min / max beside it).  Between the enqueue of a
    # window and its synchronize the device is still busy: that is when clocks / power / temperature are read.
    n_win = max(1, int(os.environ.get("OMNI_BENCH_WINDOWS", "5")))
    windows, enqueue, under_load = [], [], []
    state0 = guard.state.clone()
    first_out = None
    for w_i in range(n_win):
        _sync()
        if world > 1:
            dist.barrier()
        _sync()
        t0 = time.perf_counter()
        for s_i in range(args.steps):
            step()
            if w_i == 0 and s_i == 0:
                first_out = guard.out.clone()          # (one small copy in the whole timed region: the first step's loss)
        enqueue.append(time.perf_counter() - t0)   # host time to enqueue the K steps (diagnostic: < dt means GPU-bound)
        if gpu:
            under_load.append(gpu.sample())
        _sync()
        if world > 1:
            dist.barrier()
        _sync()
        w_dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([w_dt], dtype=torch.float64, device=DEVICE)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w_dt = float(t.item())
        windows.append(w_dt)
    order = sorted(range(n_win), key=lambda i: windows[i])
    mid = order[(n_win - 1) // 2]                  # the median window (lower median for an even count)
    dt, t_enqueue = windows[mid], enqueue[mid]
    state["under_load"] = under_load[mid] if under_load else None
    state["under_load_all_windows"] = under_load
    state["neighbours"] = gpu.neighbours() if gpu else None
    state["pci_address"] = gpu.address if gpu else None
    # one more window of the same length WITH the per-stage device timestamps (end of every critical-path graph M_k and of every
    # weight-gradient graph W_k after the step's first launch): never part of the measured windows
    stage_ends = None
    if DEVICE == "cuda" and getattr(graphed, "stages", None) and os.environ.get("OMNI_BENCH_SKIP_STAGE_ENDS") != "1":
        from omni3d_amd.cubercnn.solver import graphed as _gm
        was = _gm._PIPE_TIMING
        _gm.set_pipe_timing(True)
        _sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        _sync()
        t_win = time.perf_counter() - t1
        tab = _gm.pipe_timing_table(graphed, last=args.steps)
        _gm.set_pipe_timing(was)
        if tab:
            stage_ends = {"M_end_ms": [None if v is None else round(v, 3) for v in tab["M_end_ms"]],
                          "W_end_ms": [None if v is None else round(v, 3) for v in tab["W_end_ms"]],
                          "window_ms_per_step": round(1e3 * t_win / args.steps, 4)}
    if os.environ.get("OMNI_PIPE_TIMING") == "1" and getattr(graphed, "_timing", None):
        import sys
        from omni3d_amd.cubercnn.solver.graphed import pipe_timing_report
        print("pipe timing: " + pipe_timing_report(graphed), file=sys.stderr)
    exchange = None
    if world > 1:        # what a bad scaling curve would have to be explained with
        from omni3d_amd.cubercnn.solver.graphed import pipe_timing_table
        exchange = opt.exchange_report() or {}
        exchange["stage_timeline"] = pipe_timing_table(graphed) if getattr(graphed, "_timing", None) else None
        exchange["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "OMNI_EXCHANGE", "TORCH_NCCL"))}
    final_losses = [float(first_out[2]) if first_out is not None else float("nan"), float(guard.out[2])]
    skipped_steps = int(round(float(guard.state[2] - state0[2])))
    ims = IMS_PER_GPU * world * args.steps / dt
    step_tf = TRAIN_GFLOP_PER_IMAGE * 1e9 * IMS_PER_GPU * args.steps / dt / 1e12   # per GPU (the per-image figure is for 512 x 512)
    res = {
        "metric": f"images/sec train {MODEL_NAME} b={IMS_PER_GPU}/GPU", "value": ims, "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cubercnn_{MODEL_NAME} train step, batch {IMS_PER_GPU}/GPU, synthetic Omni3D {IMAGE_SIZE}x{IMAGE_SIZE} (8 GT/img), "
                               "fwd+10 losses+bwd+allreduce+divergence guard+SGD, random-init weights",
                   "global_batch": IMS_PER_GPU * world, "image": f"{IMAGE_SIZE}x{IMAGE_SIZE}",
                   "parallelism": f"dp{world} (flat-bucket RCCL all-reduce)"},
        "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps,
        "windows": {"ms_per_step": [round(1e3 * w / args.steps, 4) for w in windows], "min": 1e3 * min(windows) / args.steps,
                    "median": 1e3 * dt / args.steps, "max": 1e3 * max(windows) / args.steps,
                    "value_is": f"median of {n_win} back-to-back windows of {args.steps} steps, each bracketed by synchronize + barrier, MAX over ranks",
                    "conditioning_steps_before_warmup": n_cond},
        "gpu_state": state,
        "stage_ends": stage_ends,
        "launch_mode": graph_note,
        "step_mfma_frac": step_tf / FP32_MFMA_PEAK_TF,
        "step_algorithmic_tflops_per_gpu": step_tf,
        # the flops the MFMA kernels really execute (Winograd layers at the size of their point GEMMs: 2.25x / 4x fewer than the
        # direct-convolution count above) -- this, not the algorithmic figure, is the hardware's MFMA utilisation over the whole step
        "step_executed_gflop": executed_flops / 1e9,
        "step_executed_tflops_per_gpu": executed_flops * args.steps / dt / 1e12,
        "step_executed_mfma_frac": executed_flops * args.steps / dt / 1e12 / FP32_MFMA_PEAK_TF,
        "loss_first_last": [final_losses[0], final_losses[-1]],
        "skipped_steps": skipped_steps,
        "guard": "rolling-loss divergence test + NaN/Inf gradient scan + retry decision on the device, one 12-float all-reduce/step",
    }
    if exchange is not None:
        res["exchange"] = exchange
    if NONSTANDARD:
        res["nonstandard"] = {"ims_per_gpu": IMS_PER_GPU, "image_size": IMAGE_SIZE, "overrides": OVERRIDES, "device": DEVICE,
                              "note": "not BASELINE.json's configuration: functional check only"}
    if rank == 0 and DEVICE != "cuda":
        res["roofline"], res["cpu_baseline"] = None, None
    elif rank == 0 and os.environ.get("OMNI_BENCH_SKIP_ROOFLINE") == "1":      # A/B sweeps: the step time only
        res["roofline"], res["cpu_baseline"] = None, None
    elif rank == 0:
        res["roofline"] = dominant_kernel_roofline()
        res["hbm_bound_kernels"] = hbm_bound_kernels(opt)
        try:
            res["bf16_split"] = bf16_split_experiment()
        except Exception as e:  # noqa: BLE001 -- an experiment must never take the measured line down with it
            res["bf16_split"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        if world == 1 and os.environ.get("OMNI_BENCH_SKIP_DROPIN") != "1":
            # the drop-in loop (north_star: "tools/train_net.py drops in unchanged"): same step, reached from inside model(data)
            def _clean():        # every leg builds its own model + captured steps: hand their memory back before the next one starts
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            graphed, step = None, None
            _clean()
            try:
                d1 = dropin_loop(30, 1)
                _clean()
                d10 = dropin_loop(30, 10)
                _clean()
                res["dropin_loop_ms_per_step"] = d1["ms_per_step"]
                try:
                    res["dropin_loop_multiscale"] = dropin_loop_multiscale(fixed_ms=1e3 * dt / args.steps)
                except Exception as e:  # noqa: BLE001
                    res["dropin_loop_multiscale"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                _clean()
                if os.environ.get("OMNI_BENCH_SKIP_STREAM") != "1":
                    try:
                        res["dropin_loop_multiscale_stream"] = dropin_loop_multiscale_stream(fixed_ms=1e3 * dt / args.steps)
                    except Exception as e:  # noqa: BLE001
                        res["dropin_loop_multiscale_stream"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                res["dropin_loop"] = {"loop": "tools/train_synthetic.py == tools/train_net.py:176-285 body, new batch (pool of 4) every iteration, "
                                              "host-side target packing + H2D inside the timed region",
                                      "losses_read_every_iteration": d1, "losses_read_every_10th": d10,
                                      "vs_ms_per_step": d1["ms_per_step"] / (1e3 * dt / args.steps)}
            except Exception as e:  # noqa: BLE001
                res["dropin_loop_ms_per_step"] = None
                res["dropin_loop"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1:   # the CPU leg is reported at N = 1 only (a minute of host work the other ranks would wait on)
            res["cpu_baseline"] = cpu_baseline_train(priors)
        else:
            res["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": "reported at N=1 only"}
    return res


def dropin_loop(iters=30, sync_every=1):
    """The reference's loop body (tools/train_net.py:176-285; tools/train_synthetic.py is the same loop as a script) on a FRESH model:
    loss_dict = model(data) -> sum -> optimizer.zero_grad() -> backward() -> non-finite scan -> guard -> optimizer.step() ->
    scheduler.step(), a new batch from a pool of four every iteration.  Nothing here knows about graphs: `model(data)` and the
    optimizer switch to the staged-graph replay by themselves once the batch signature has repeated (cubercnn/solver/
    autoreplay.py).  sync_every = 1 reads the reduced losses back on every iteration like the reference (:186); larger values
    read them every n-th iteration.  -> ms per iteration over the replayed iterations, number of replayed iterations"""
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    from omni3d_amd.d2.solver import build_lr_scheduler
    cfg, model, opt, priors = build(1, seed=1)
    sched = build_lr_scheduler(cfg, opt)
    auto = model.__dict__.get("_omni_auto")
    pool = [synthetic.make_batch(IMS_PER_GPU, IMAGE_SIZE, IMAGE_SIZE, num_gt=8, seed=2000 + s, priors=priors) for s in range(4)]
    guard = None
    warm = (auto.warm + 1) if auto is not None else 1

    def iteration(it):
        nonlocal guard
        loss_dict = model(pool[it % len(pool)])
        losses = sum(loss_dict.values())
        if guard is None:
            guard = StepGuard(list(loss_dict), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, losses.device)
            opt.skip_flag = guard.skip
        opt.zero_grad()
        losses.backward()
        opt.all_reduce_grads()
        opt.check_nonfinite(guard.nonfinite_flag)
        guard.update(loss_dict, sync=(it % sync_every == 0))
        opt.step()
        sched.step()

    for it in range(warm + 2):              # eager warm-up iterations, the capture, first replays
        iteration(it)
    _sync()
    t0 = time.perf_counter()
    for it in range(warm + 2, warm + 2 + iters):
        iteration(it)
    _sync()
    dt = time.perf_counter() - t0
    skipped, retry, red = guard.read()
    return {"ms_per_step": 1e3 * dt / iters, "iterations": iters, "replayed_iterations": auto.replays if auto is not None else 0,
            "capture": "ok" if (auto is not None and auto.failed is None and auto.stepper is not None) else f"not replaying: {getattr(auto, 'failed', 'disabled')}",
            "losses_read_every": sync_every, "last_total_loss": red["total_loss"]}


def dropin_loop_multiscale(iters=24, pool_size=8, fixed_ms=None):
    """The same loop body on batches drawn like the reference's loader draws them (configs/Base.yaml:10-13: a new short edge per
    image, cubercnn/data/dataset_mapper.py:17-58; synthetic.make_multiscale_batch): every batch has its own tuple of image shapes.
    AutoReplay keys its captured steps by (batch size, padded height, padded width) -- ImageList pads to multiples of 64 anyway -- and
    serves any image sizes of a bucket from one capture (images in masked slots, sizes as device data).  Reported: ms per iteration
    once every bucket of the pool is captured, the buckets, and the fixed-shape replayed step scaled by padded pixels for comparison."""
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    from omni3d_amd.d2.solver import build_lr_scheduler
    cfg, model, opt, priors = build(1, seed=1)
    sched = build_lr_scheduler(cfg, opt)
    auto = model.__dict__.get("_omni_auto")
    if auto is None:
        return {"error": "AutoReplay disabled"}
    pool = [synthetic.make_multiscale_batch(IMS_PER_GPU, 3000 + s, priors=priors) for s in range(pool_size)]
    sigs = [auto.signature(b, record=False) for b in pool]
    guard = None

    def iteration(it):
        nonlocal guard
        loss_dict = model(pool[it % len(pool)])
        losses = sum(loss_dict.values())
        if guard is None:
            guard = StepGuard(list(loss_dict), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, losses.device)
            opt.skip_flag = guard.skip
        opt.zero_grad()
        losses.backward()
        opt.all_reduce_grads()
        opt.check_nonfinite(guard.nonfinite_flag)
        guard.update(loss_dict, sync=True)
        opt.step()
        sched.step()

    it = 0
    for _ in range((auto.warm + 2) * len(pool)):     # every bucket: its eager warm-up iterations, its capture, a first replay
        iteration(it)
        it += 1
    _sync()
    replays0, captures0 = auto.replays, auto.captures
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration(it)
        it += 1
    _sync()
    dt = time.perf_counter() - t0
    px = [s[1] * s[2] for s in sigs]
    mean_px = sum(px[(k + (auto.warm + 2) * len(pool)) % len(pool)] for k in range(iters)) / iters
    out = {"ms_per_step": 1e3 * dt / iters, "iterations": iters, "pool": pool_size, "buckets": sorted(set(sigs)),
           "image_shapes_of_first_batch": [tuple(b["image"].shape[-2:]) for b in pool[0]],
           "replayed_in_timed_region": auto.replays - replays0, "captures_in_timed_region": auto.captures - captures0,
           "captures_total": auto.captures, "capture": "ok" if auto.failed is None else f"not replaying: {auto.failed}",
           "mean_padded_pixels_per_image": mean_px}
    if fixed_ms is not None:
        out["fixed_shape_step_scaled_by_pixels_ms"] = fixed_ms * mean_px / float(IMAGE_SIZE * IMAGE_SIZE)
        out["vs_pixel_scaled_fixed_shape"] = out["ms_per_step"] / out["fixed_shape_step_scaled_by_pixels_ms"]
    return out


def dropin_loop_multiscale_stream(iters=320, fixed_ms=None):
    """the same loop on a NON-RECYCLED stream -- `iters` freshly drawn batches (generated before the clock starts,
    never repeated), a cold cache: the timed region contains the eager warm-up iterations of every new size bucket, its capture, the
    evictions of the memory-bounded 64-entry cache and whatever the thrash guard decides (cubercnn/solver/autoreplay.py).  Reported: the cache's own
    statistics, the time per iteration over the WHOLE region and over its last quarter (steady state)."""
    from omni3d_amd import synthetic
    from omni3d_amd.cubercnn.solver.guard import StepGuard
    from omni3d_amd.d2.solver import build_lr_scheduler
    import warnings
    cfg, model, opt, priors = build(1, seed=2)
    sched = build_lr_scheduler(cfg, opt)
    auto = model.__dict__.get("_omni_auto")
    if auto is None:
        return {"error": "AutoReplay disabled"}
    stream = [synthetic.make_multiscale_batch(IMS_PER_GPU, 7000 + s, priors=priors) for s in range(iters)]
    guard = None
    marks = []
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        _sync()
        t0 = time.perf_counter()
        kinds, t_prev = {"eager": [], "capture": [], "replay": []}, time.perf_counter()
        slow = []
        for it, batch in enumerate(stream):
            cap0, rep0 = auto.captures, auto.replays
            r0 = torch.cuda.memory_reserved() if DEVICE == "cuda" else 0
            loss_dict = model(batch)
            losses = sum(loss_dict.values())
            if guard is None:
                guard = StepGuard(list(loss_dict), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, losses.device)
                opt.skip_flag = guard.skip
            opt.zero_grad()
            losses.backward()
            opt.all_reduce_grads()
            opt.check_nonfinite(guard.nonfinite_flag)
            guard.update(loss_dict, sync=True)
            opt.step()
            sched.step()
            # (per-iteration wall time by kind: the loss read-back above is the iteration's synchronisation point)
            t_now = time.perf_counter()
            kind = "capture" if auto.captures > cap0 else "replay" if auto.replays > rep0 else "eager"
            kinds[kind].append(1e3 * (t_now - t_prev))
            # (diagnostic: the iterations that took > 150 ms with what the caching allocator did during them -- a run whose region is slow
            # says which kind of iteration and whether memory was bought from / handed back to the driver)
            if 1e3 * (t_now - t_prev) > 150.0:
                slow.append([it, kind, round(1e3 * (t_now - t_prev), 1), round(((torch.cuda.memory_reserved() if DEVICE == "cuda" else 0) - r0) / 2 ** 20)])
            t_prev = t_now
            if it + 1 == iters - iters // 4:
                _sync()
                marks.append((time.perf_counter(), auto.replays))
                t_prev = time.perf_counter()
        _sync()
        t1 = time.perf_counter()
    st = auto.stats()
    px = [s_[1] * s_[2] for s_ in (auto.signature(b, record=False) for b in stream)]
    out = {"iterations": iters, "ms_per_iteration_whole_region": 1e3 * (t1 - t0) / iters,
           "ms_per_iteration_last_quarter": 1e3 * (t1 - marks[0][0]) / (iters // 4) if marks else None,
           "replayed_in_last_quarter": (auto.replays - marks[0][1]) if marks else None, "last_quarter_iterations": iters // 4,
           "cache_entries": len(auto.cache), "stats": st, "capture": "ok" if auto.failed is None else f"not replaying: {auto.failed}",
           "guard_warnings": [str(r.message)[:160] for r in rec if "omni3d_amd" in str(r.message)],
           "mean_padded_pixels_per_image": sum(px) / len(px),
           "slow_iterations_it_kind_ms_reservedMiB": slow[:24], "slow_iterations_total_ms": round(sum(r[2] for r in slow), 1)}

    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if v else None
    out["hit_rate"] = st.get("hit_rate")
    out["eager_new_shape_ms"] = med(kinds["eager"])          # an iteration on a bucket that is not captured yet (eager launches)
    out["capture_ms"] = med(kinds["capture"])
    out["replay_ms"] = med(kinds["replay"])
    out["iterations_by_kind"] = {k: len(v) for k, v in kinds.items()}
    if fixed_ms is not None:
        out["fixed_shape_step_scaled_by_pixels_ms"] = fixed_ms * out["mean_padded_pixels_per_image"] / float(IMAGE_SIZE * IMAGE_SIZE)
        out["whole_region_vs_pixel_scaled_fixed_shape"] = out["ms_per_iteration_whole_region"] / out["fixed_shape_step_scaled_by_pixels_ms"]
    return out


def run_infer(args, world, rank):
    """`--workload infer`: RCNN3D.inference (backbone, RPN, box head, per-class NMS, cube head decode) on the same staged batch;
    not the BASELINE metric, reported for the inference half of the path.  Images shard across ranks, no collective."""
    cfg, model, opt, priors = build(world)
    batch, _ = stage_batch(model, priors, rank)
    model.eval()
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            out = model(batch)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(batch)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"metric": f"images/sec inference {MODEL_NAME} b=4/GPU", "value": IMS_PER_GPU * world * args.steps / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cubercnn_{MODEL_NAME} inference, batch 4/GPU, synthetic Omni3D 512x512, random-init weights "
                                   "(score threshold 0.01 keeps ~all 100 detections/image)",
                       "detections_first_image": int(len(out[0]["instances"]))}}


def _time_launch(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # torch's current stream is
    torch.cuda.synchronize()                                                            # the stream the launchers use
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def _latest(name):
    """newest committed round of a profile artefact (profiles/r05_<name>, else r04_<name>, ...)"""
    from omni3d_amd.profile_io import latest_profile
    return latest_profile(name)


def _trace_table_rows(name=None):
    """rows of the committed per-(kernel, grid) table of ONE replayed step (tools/trace_table.py): {(symbol, grid): (launches per step,
    average us)} -- what a kernel takes INSIDE the step, beside the other stream's work, as opposed to the isolated timings"""
    import re
    name = name or _latest("trace_table_final.txt")
    path = os.path.join(ROOT, "profiles", name)
    rows = {}
    if not os.path.exists(path):
        return rows, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+(\d+)x(\d+)x(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows[(m.group(1).strip(), f"{m.group(2)}x{m.group(3)}x{m.group(4)}")] = (float(m.group(5)), float(m.group(6)))
    return rows, f"profiles/{name}"


def _table_shares(csv_name=None):
    """share of the summed kernel time per kernel symbol in the committed rocprofv3 table (+ its sha256) -> ({symbol: frac}, note)"""
    import csv
    import hashlib
    csv_name = csv_name or _latest("train_final_kernel_stats.csv")
    path = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(path):
        return {}, None
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    def short(n):       # "void (anonymous namespace)::conv_fwd_kernel<64, 64, 2, 2, 32>((anonymous namespace)::ConvP)" -> symbol
        n = n[5:] if n.startswith("void ") else n
        n = n[len("(anonymous namespace)::"):] if n.startswith("(anonymous namespace)::") else n
        return n.split("(")[0].strip()
    shares = {}
    for r in rows:
        shares[short(r["Name"])] = shares.get(short(r["Name"]), 0.0) + float(r["TotalDurationNs"]) / tot
    return shares, f"profiles/{csv_name} sha256:{hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]}"


def dominant_kernel_roofline(iters=20):
    """`roofline`: the kernel symbol that tops the committed rocprofv3 table of this command (profiles/r03_train_final_kernel_stats.csv),
    on the launch shape that accounts for most of that symbol's time, timed live with HIP events on the launch stream; HBM traffic and
    the cycle-based MFMA utilisation come from the committed PMC summary (`traffic_source`).  `families`: every MFMA kernel family of
    the table with one representative launch timed the same way and its share of the summed kernel time, so the per-family distance
    to the 157.3 TFLOP/s fp32-MFMA peak is in the line (the step is a near tie between six MFMA symbols at 7-9 % each)."""
    from omni3d_amd.kernels import conv, wino
    from omni3d_amd.profile_io import profile_counters
    PMC = _latest("pmc_families.csv")
    shares, table = _table_shares()
    B, C, H = IMS_PER_GPU, 256, 128
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
    V, U = wino.transform_input(x, 4), wino.transform_weights(w, tile=4)[0]
    P, T = V.shape[0], V.shape[1]
    flops = 2.0 * P * T * C * C
    ms_direct = _time_launch(lambda: conv.conv2d_fwd(x, w, None, 1, 1), iters)
    flops_direct = 2.0 * B * H * H * C * C * 9
    ms_wino = _time_launch(lambda: wino.conv3x3_fwd(x, w, tile=4), iters)

    step_rows, step_src = _trace_table_rows()

    def fam(name, kernel, shape, fl, fn, pmc_key=None, grid=None, alg_bytes=None, per_step=None, step_grid=None, step_gflop=None):
        t = _time_launch(fn, max(iters // 2, 5))
        sym = kernel.split("(")[0]
        r = {"family": name, "kernel": kernel, "shape": shape, "gflop": fl / 1e9, "kernel_ms": t, "tflops": fl / (t * 1e-3) / 1e12,
             "frac": fl / (t * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, "share_of_kernel_time_in_table": shares.get(sym)}
        if per_step is not None:
            r["launches_per_step"] = per_step
        # the same (kernel, grid) INSIDE the step, beside the other stream's work: average duration in the committed table of one
        # replayed step (in-step fractions next to the isolated ones).  step_gflop: flops of the AVERAGE launch of
        # that row when it mixes shapes (the two fc1 weight gradients share a grid)
        row = None if step_grid is False else step_rows.get((sym[:60], step_grid if step_grid is not None else f"{grid}x1x1"))
        if row is not None:
            gf = step_gflop if step_gflop is not None else fl / 1e9
            r["in_step"] = {"launches_per_step": row[0], "avg_us": row[1], "tflops": gf * 1e9 / (row[1] * 1e-6) / 1e12,
                            "frac": gf * 1e9 / (row[1] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, "source": step_src}
        if alg_bytes is not None:
            r["algorithmic_bytes_per_launch"] = alg_bytes
        c = profile_counters(PMC, pmc_key or kernel.split("<")[0], grid)
        if c and c.get("mfma_busy_frac"):
            r["pmc_mfma_busy_frac"], r["pmc_source"] = c["mfma_busy_frac"], f"{c['source']} sha256:{c['sha256_16']} grid {c['grid']}"
        if c and c.get("FETCH_SIZE_x2_MB") and c.get("WRITE_SIZE_MB") and grid is not None:
            r["pmc_traffic_bytes"] = (c["FETCH_SIZE_x2_MB"] + c["WRITE_SIZE_MB"]) * 1e6
        return r
    dM = torch.randn(P, T, C, device="cuda")
    x1, w1 = torch.randn(2048, 12544, device="cuda"), torch.randn(1024, 12544, device="cuda") * 0.02
    dy1 = torch.randn(2048, 1024, device="cuda")
    gacc1 = torch.zeros(1024, 12544, device="cuda")
    xs = torch.randn(B, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    ws = (torch.randn(128, 64, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    dys = torch.randn(B, 128, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    gws = torch.zeros(128, 64, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    dyl1 = torch.randn(B, 32, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
    wl1 = (torch.randn(32, 16, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    x3 = torch.randn(B, 128, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    w3 = (torch.randn(128, 128, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    V3, U3 = wino.transform_input(x3, 4), wino.transform_weights(w3, tile=4)[0]           # 36 x [1024 x 128], 36 x [128 x 128]
    dM3 = torch.randn(36, 1024, 128, device="cuda")
    fl3 = 2.0 * 36 * 1024 * 128 * 128
    V4, U4 = torch.randn(36, 256, 256, device="cuda"), torch.randn(36, 256, 256, device="cuda")      # DLA level 4: 256 tiles of 4x4 at batch 4
    dM4 = torch.randn(36, 256, 256, device="cuda")
    V5 = torch.randn(36, 1024, 256, device="cuda")                                                   # FPN output / RPN conv at p3
    fl4 = 2.0 * 36 * 256 * 256 * 256
    # the weight-gradient stream's launch of backward stage 2 (FPN output convolutions p2..p5, DLA level 5 and level 4): 14 problems
    STAGE2 = [(36, 4096, 256, 256), (36, 1024, 256, 256), (36, 256, 256, 256), (16, 256, 256, 256)] + [(16, 256, 512, 512)] * 3 + [(36, 256, 256, 256)] * 7
    multi = [(torch.randn(b, m, c, device="cuda"), torch.randn(b, m, k, device="cuda")) for b, m, k, c in STAGE2]
    fl_multi = sum(2.0 * b * m * k * c for b, m, k, c in STAGE2)
    families = [
        fam("Winograd weight-gradient GEMMs of one backward stage in ONE launch (round 4: 14 problems -- FPN p2..p5, DLA level 5, level 4)",
            "gemm_tn_multi_kernel<4>", "36x[4096,256,256] + 36x[1024,256,256] + 16x[256,256,256] + 3 x 16x[256,512,512] + 8 x 36x[256,256,256]",
            fl_multi, lambda: wino.gemm_batched_wgrad_multi(multi), grid=2326528),
        fam("Winograd point GEMMs, 128x128 maps (FPN output / RPN conv at p2: 4 launches / step, the heaviest shape of this symbol)", "gemm_nt_pf_kernel<4>", "36x[4096x256]x[256x256]^T (3x3 256->256 @128x128)", flops,
            lambda: wino.gemm_batched(V, U), grid=2359296, alg_bytes=4.0 * (2 * P * T * C + P * C * C), per_step=4),
        fam("Winograd point GEMMs, small maps (DLA level 4: 18 launches / step)", "gemm_nt_pf_kernel<4>",
            "36x[256x256]x[256x256]^T (3x3 256->256 @32x32, F(4x4,3x3))", fl4, lambda: wino.gemm_batched(V4, U4), grid=147456,
            alg_bytes=4.0 * (2 * 36 * 256 * 256 + 36 * 256 * 256), per_step=18),
        fam("Winograd point GEMMs, small maps (DLA level 3: 14 launches / step)", "gemm_nt_pf_kernel<4>",
            "36x[1024x128]x[128x128]^T (3x3 128->128 @64x64, F(4x4,3x3))", fl3, lambda: wino.gemm_batched(V3, U3), grid=294912,
            alg_bytes=4.0 * (2 * 36 * 1024 * 128 + 36 * 128 * 128), per_step=14),
        fam("Winograd point GEMMs, 64x64 maps (FPN / RPN p3)", "gemm_nt_pf_kernel<4>",
            "36x[1024x256]x[256x256]^T (3x3 256->256 @64x64, F(4x4,3x3))", 2.0 * 36 * 1024 * 256 * 256,
            lambda: wino.gemm_batched(V5, U4), grid=589824, alg_bytes=4.0 * (2 * 36 * 1024 * 256 + 36 * 256 * 256), per_step=4),
        fam("Winograd weight-gradient GEMMs", "gemm_tn_pf_kernel<4>", "36x[256x4096]x[4096x256] (same layer)", flops,
            lambda: wino.gemm_batched_wgrad(V, dM), grid=147456, step_grid=False),      # (PMC: the three weight-gradient shapes share the geometry
                                                                        #  576 workgroups -> one averaged row)
        fam("FC forward (fc1-class)", "gemm_engine_kernel<0, 0, 128, 128, false>", "[2048x12544]x[1024x12544]^T box-head fc1 (128 tiles x 2 reduction halves = one round of 256 workgroups)",
            2.0 * 2048 * 12544 * 1024, lambda: conv.linear_fwd(x1, w1, None), pmc_key="gemm_engine_kernel<0, 0, 128, 128, false>", grid=65536),
        fam("FC data gradient (the engine's NN form reading W as it is, balanced work split)", "gemm_engine_kernel<0, 1, 128, 128, true>",
            "[2048x1024]x[1024x12544] box-head fc1 (1568 tiles = 6 per workgroup + 32 tiles cut in 8)",
            2.0 * 2048 * 12544 * 1024, lambda: conv.linear_dgrad(dy1, w1), pmc_key="gemm_engine_kernel<0, 1, 128, 128, true>", grid=65536,
            step_gflop=(2.0 * (2048 + 512) * 12544 * 1024) / 2 / 1e9),       # (in the step: the box head's and the cube head's launch, averaged)
        fam("FC weight gradient (round 5: a tile kernel on the weight-gradient stream -- the engine's balanced form held 410 registers "
            "per lane on every CU and stalled the main stream, kernels/conv.py; late round 6: 64x64 tiles, 3136 of them end together where "
            "1568 of 128x64 left a third round of 32 workgroups; accumulated into the gradient bucket; in the step the row averages "
            "the box head's 2048-row and the cube head's 512-row launch; deep-prefetch operand path since late round 6)", "conv_wgrad_pf_kernel<64, 64, 2, 2, 32, 2>",
            "[1024x2048]x[2048x12544] box-head fc1 (16 x 196 tiles)", 2.0 * 2048 * 12544 * 1024,
            lambda: conv.linear_wgrad(x1, dy1, accum_into=gacc1), grid=802816, per_step=2, step_gflop=(2.0 * (2048 + 512) * 12544 * 1024) / 2 / 1e9),
        fam("Winograd weight-gradient GEMMs, small maps", "gemm_tn_pf_kernel<4>", "36x[128x1024]x[1024x128] (DLA level 3)", fl3,
            lambda: wino.gemm_batched_wgrad(V3, dM3), grid=147456, step_grid=False),
        fam("Winograd weight-gradient GEMMs, small maps (DLA level 4)", "gemm_tn_pf_kernel<4>", "36x[256x256]x[256x256] (DLA level 4)", fl4,
            lambda: wino.gemm_batched_wgrad(V4, dM4), grid=147456, step_grid=False),
        fam("direct conv 64x64 tiles (23 launches / step: the stride-2 3x3 and the 1x1 root / projection / lateral layers)", "conv_fwd_kernel<64, 64, 2, 2, 32, 1>",
            "3x3/s2 64->128 @128x128 (DLA level 3 entry)", 2.0 * B * 64 * 64 * 128 * 64 * 9,
            lambda: conv.conv2d_fwd(xs, ws, None, 2, 1), grid=131072, per_step=23, step_grid=False),     # (six shapes share this grid in the step)
        fam("stride-2 data gradient, all four parity classes of a dx tile in one workgroup (round 6, csrc/dgrad_s2.hip; the generic kernel's "
            "grid.z classes moved 79.9 MB for these 25.5 MB of operands)", "dgrad_s2_kernel<64>", "3x3/s2 64->128 @128x128",
            2.0 * B * 64 * 64 * 128 * 64 * 9, lambda: conv.conv2d_dgrad(dys, ws, (128, 128), 2, 1), grid=65536, step_grid="65536x1x1", per_step=1,
            alg_bytes=4.0 * (B * 64 * 64 * 128 + B * 128 * 128 * 64 + 128 * 9 * 64)),
        fam("direct wgrad 128x64 tiles (VERDICT r4 missing 2: the #2 symbol of the round-4 table; 23 launches / step on the weight-gradient stream)",
            "conv_wgrad_pf_kernel<128, 64, 2, 2, 32, 2>", "3x3/s2 64->128 @128x128: [128 x 65536]x[65536 x 576], 9 tiles x 64 pixel splits",
            2.0 * B * 64 * 64 * 128 * 64 * 9, lambda: conv.conv2d_wgrad(xs, dys, (3, 3), 2, 1, accum_into=gws), grid=2304 * 64, step_grid="2304x64x1", per_step=23),
        fam("stem data gradient, stride 2 (round 5, MFMA 16x16x4: 2.4 GFLOP against 100 MB -- HBM floor 12.5 us, MFMA floor 15 us)", "stem_dgrad_s2_kernel<16, 32>",
            "3x3/s2 16->32 @512x512 (DLA level1), dx from dy", 2.0 * B * 256 * 256 * 32 * 16 * 9,
            lambda: conv.stem_conv_dgrad(dyl1, wl1, (512, 512), 2), grid=524288, alg_bytes=4.0 * B * (256 * 256 * 32 + 512 * 512 * 16), per_step=1),
        fam("direct conv 128x128 tiles", "conv_fwd_kernel<128, 128, 2, 2, 32, 1>", "3x3 256->256 @128x128 (the same layer WITHOUT Winograd)", flops_direct,
            lambda: conv.conv2d_fwd(x, w, None, 1, 1), pmc_key="(not in the production dispatch: no PMC row)"),
        fam("direct conv mid layers", "conv_fwd_kernel<64, 64, 2, 2, 32, 1>", "3x3 128->128 @64x64 (DLA level 3 block, direct)", 2.0 * B * 64 * 64 * 128 * 128 * 9,
            lambda: conv.conv2d_fwd(x3, w3, None, 1, 1), pmc_key="(not in the production dispatch: no PMC row)"),
    ]
    # headline = the family whose kernel symbol tops the committed table (first family of that symbol in the list above)
    top = max(shares, key=shares.get) if shares else None
    head = next((f for f in families if top is not None and f["kernel"].split("(")[0] == top), families[0])
    traffic = head.get("pmc_traffic_bytes")
    # `frac` is the symbol's BEST (and heaviest) shape; the same symbol over all of its shapes of a step, weighted
    # by launches per step, and the launch-weighted mean over every family listed below
    same = [f for f in families if f["kernel"] == head["kernel"] and f.get("launches_per_step")]
    sym_w = (sum(f["gflop"] * f["launches_per_step"] for f in same) / max(sum(f["kernel_ms"] * f["launches_per_step"] for f in same), 1e-9)
             / FP32_MFMA_PEAK_TF) if same else None
    return {"bound": "mfma",
            "frac_symbol_weighted": sym_w,
            "frac_symbol_weighted_note": "same kernel symbol over %d launches / step of %d shapes (flops x launches / time x launches of the "
                                         "families below that carry `launches_per_step`); `frac` is its heaviest shape" % (
                                             sum(f["launches_per_step"] for f in same), len(same)) if same else None,
            "kernel": f"{head['kernel']} on {head['shape']} [{head['family']}]"
                      + (f" -- tops {table} with {shares[top]:.1%} of the summed kernel time" if top == head["kernel"].split("(")[0] else ""),
            "achieved": head["tflops"], "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": head["frac"],
            "traffic": traffic,
            "traffic_source": head.get("pmc_source", None) and (head["pmc_source"] + " (rocprofv3 --pmc passes of tools/run_families.py: 2 x FETCH_SIZE "
                                                                  "+ WRITE_SIZE, gfx950 wide-read correction of MI355X_MICROARCH.md)"),
            "pmc_mfma_busy_frac": head.get("pmc_mfma_busy_frac"),
            "algorithmic_bytes_per_launch": head.get("algorithmic_bytes_per_launch"),
            "kernel_ms": head["kernel_ms"], "flops_per_launch": head["gflop"] * 1e9, "operands": "fp32 (v_mfma_f32_32x32x2_f32)",
            "table": table, "layer_ms_winograd_vs_direct_3x3_256_at_128": [ms_wino, ms_direct],
            "families": families}


def bf16_split_experiment(iters=20):
    """VERDICT r5 item 8 / SURVEY.md section 7: the OPT-IN bf16-split form of the Winograd point GEMM (csrc/gemm_split.hip) next to the
    product's fp32-MFMA kernel on the same inputs -- time, fp32-equivalent rate and error against float64 (four of the 36 points).
    A separate object with its own `operands` strings and its own peak; the measured line never uses it."""
    from omni3d_amd.kernels import wino
    g = torch.Generator(device="cuda").manual_seed(0)
    rows = []
    for name, B, M, K, C in (("p2 point GEMM 36x[4096x256]x[256x256]^T", 36, 4096, 256, 256), ("DLA level 3 36x[1024x128]x[128x128]^T", 36, 1024, 128, 128)):
        V = torch.randn(B, M, C, device="cuda", generator=g)
        U = torch.randn(B, K, C, device="cuda", generator=g) * 0.05
        ref = torch.bmm(V[:4].double(), U[:4].double().transpose(1, 2))
        scale = float(ref.abs().max())
        fl = 2.0 * B * M * K * C
        row = {"shape": name, "gflop": fl / 1e9}
        for key, fn, operands, peak in (
                ("fp32_mfma", lambda: wino.gemm_batched(V, U), "fp32 (v_mfma_f32_32x32x2_f32)", FP32_MFMA_PEAK_TF),
                ("split6", lambda: wino.gemm_batched_split(V, U, 6), "3 bf16 planes per fp32 operand, 6 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate", 2500.0 / 6),
                ("split3", lambda: wino.gemm_batched_split(V, U, 3), "2 bf16 planes per fp32 operand, 3 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate", 2500.0 / 3)):
            out = fn()
            err = float((out[:4].double() - ref).abs().max()) / scale
            ms = _time_launch(fn, iters)
            row[key] = {"kernel_ms": ms, "tflops_fp32_equivalent": fl / ms / 1e9, "peak_fp32_equivalent": peak, "operands": operands,
                        "max_err_vs_fp64_over_max_ref": err}
        row["split6_err_over_fp32_err"] = row["split6"]["max_err_vs_fp64_over_max_ref"] / max(row["fp32_mfma"]["max_err_vs_fp64_over_max_ref"], 1e-30)
        row["split3_err_over_fp32_err"] = row["split3"]["max_err_vs_fp64_over_max_ref"] / max(row["fp32_mfma"]["max_err_vs_fp64_over_max_ref"], 1e-30)
        rows.append(row)
    return {"note": "opt-in experiment (OMNI_GEMM_SPLIT): never part of the measured step; accepted as admissible only where its error is <= the "
                    "fp32-MFMA kernel's on the same inputs", "rows": rows}


def hbm_bound_kernels(opt, iters=10):
    """The HBM-bound side of the step, reported separately: algorithmic bytes / live HIP-event time."""
    from omni3d_amd.kernels import bnpool, wino
    out = []
    n = opt.flat_param.numel()
    ms = _time_launch(lambda: opt.step(), iters)          # p, g, m read; p, m written
    out.append({"kernel": "sgd_kernel (flat bucket, 47.9 M params)", "bytes": 20.0 * n, "kernel_ms": ms})
    x = torch.randn(IMS_PER_GPU, 16, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)
    g, b = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
    ms = _time_launch(lambda: bnpool.bn_fwd(x, g, b, None, None, None, True, 1e-5, 0.1), iters)   # x read twice, y written
    out.append({"kernel": "bn_reduce + bn_finalize_fwd + bn_apply (16 ch @512x512, batch 4)", "bytes": 12.0 * x.numel(), "kernel_ms": ms})
    x = torch.randn(IMS_PER_GPU, 256, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    ms = _time_launch(lambda: wino.transform_input(x), iters)                                    # x read, 4x written
    out.append({"kernel": "wino_in_kernel (256 ch @128x128, batch 4)", "bytes": 20.0 * x.numel(), "kernel_ms": ms})
    # the fused RPN head over the five FPN levels of this workload (csrc/rpn_head.hip): forward reads t (256 ch) and writes 16 columns per
    # pixel; the data gradient reads dy (16) + t (ReLU mask) and writes 256 channels
    from omni3d_amd.kernels import det
    tn = [torch.relu(torch.randn(IMS_PER_GPU, s, s, 256, device="cuda")) for s in (128, 64, 32, 16, 8)]
    pix = sum(t.shape[0] * t.shape[1] * t.shape[2] for t in tn)
    wo, wd = torch.randn(3, 256, device="cuda") * 0.05, torch.randn(12, 256, device="cuda") * 0.05
    bo, bd = torch.zeros(3, device="cuda"), torch.zeros(12, device="cuda")
    ms = _time_launch(lambda: det.head16_fwd(tn, wo, bo, wd, bd), iters)
    out.append({"kernel": "head16_fwd_kernel (RPN objectness + deltas, p2..p6 in one launch)", "bytes": 4.0 * pix * (256 + 16), "kernel_ms": ms})
    dys = [torch.randn(t.shape[:3] + (16,), device="cuda") for t in tn]
    ms = _time_launch(lambda: det.head16_dgrad(dys, tn, wo, wd), iters)
    out.append({"kernel": "head16_dgrad_kernel (+ ReLU mask, p2..p6 in one launch)", "bytes": 4.0 * pix * (16 + 256 + 256), "kernel_ms": ms})
    for o in out:
        o["achieved"] = o["bytes"] / (o["kernel_ms"] * 1e-3) / 1e9
        o["peak"], o["unit"] = 8000.0, "GB/s"
        o["frac"] = o["achieved"] / 8000.0
    return out


def cpu_baseline_train(priors):
    """The CPU oracle (plain-PyTorch port of the reference path, oracle/model_oracle.py) timed on the host
    cores on a bounded sample: batch 2, forward + losses + backward + SGD."""
    if os.environ.get("OMNI_BENCH_SKIP_CPU") == "1":   # profiling runs: do not spend GPU-box minutes on the CPU leg
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": "skipped (OMNI_BENCH_SKIP_CPU=1)"}
    try:
        from oracle import model_oracle
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    try:
        return model_oracle.time_training(priors)
    except Exception as e:  # noqa: BLE001 -- a failing CPU leg must never take the measured line down with it
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {str(e)[:200]}"}
