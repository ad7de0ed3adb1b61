"""host time per phase of the drop-in loop (bench_train.dropin_loop body), averaged over replayed iterations"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from omni3d_amd import bench_train as BT, synthetic
from omni3d_amd.cubercnn.solver.guard import StepGuard
from omni3d_amd.d2.solver import build_lr_scheduler
cfg, model, opt, priors = BT.build(1, seed=1)
sched = build_lr_scheduler(cfg, opt)
auto = model.__dict__.get("_omni_auto")
pool = [synthetic.make_batch(BT.IMS_PER_GPU, BT.IMAGE_SIZE, BT.IMAGE_SIZE, num_gt=8, seed=2000 + s, priors=priors) for s in range(4)]
guard = None
T = {}
SYNC = int(os.environ.get("SYNC_EVERY", "1"))
def tick(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0.0) + t - t0; return t
def iteration(it, rec):
    global guard
    t = time.perf_counter()
    loss_dict = model(pool[it % len(pool)]); t = tick("model(data)", t) if rec else time.perf_counter()
    losses = sum(loss_dict.values()); t = tick("sum", t) if rec else time.perf_counter()
    if guard is None:
        guard = StepGuard(list(loss_dict), cfg.MODEL.STABILIZE, cfg.SOLVER.CHECKPOINT_PERIOD, losses.device); opt.skip_flag = guard.skip
    opt.zero_grad(); t = tick("zero_grad", t) if rec else time.perf_counter()
    losses.backward(); t = tick("backward", t) if rec else time.perf_counter()
    opt.all_reduce_grads(); opt.check_nonfinite(guard.nonfinite_flag); t = tick("allreduce+scan", t) if rec else time.perf_counter()
    guard.update(loss_dict, sync=(it % SYNC == 0)); t = tick("guard.update (+sync)", t) if rec else time.perf_counter()
    opt.step(); t = tick("opt.step", t) if rec else time.perf_counter()
    sched.step(); t = tick("sched.step", t) if rec else time.perf_counter()
for it in range(6):
    iteration(it, False)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for it in range(6, 6 + N):
    iteration(it, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"sync every {SYNC}: {1e3 * dt / N:.3f} ms per iteration; host ms per phase:", {k: round(1e3 * v / N, 3) for k, v in T.items()}, "sum", round(1e3 * sum(T.values()) / N, 3))
# inside model(data)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(100, 120):
    iteration(it, False)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
