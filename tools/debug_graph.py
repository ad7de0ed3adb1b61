import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omni3d_amd import bench_train as BT
from omni3d_amd.cubercnn.solver.graphed import GraphedForwardBackward

cfg, model, opt, priors = BT.build(1)
batch, packed = BT.stage_batch(model, priors, 0)


def eager():
    opt.zero_grad()
    losses = model(batch, packed)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    print("  logs", {k: [round(x, 3) for x in v.flatten().float().tolist()][:12] if torch.is_tensor(v) else str(type(v)) for k, v in model.roi_heads.pending_logs.items()},
          {k: [round(x, 3) for x in v[0].flatten().tolist()] for k, v in model.proposal_generator.pending_logs.items()}, flush=True)
    print("eager", {k: round(float(v.detach()), 4) for k, v in losses.items()}, "gnorm", float(opt.flat_grad.norm()), flush=True)


eager()
g = GraphedForwardBackward(model, opt, batch, packed)
for it in range(3):
    l, t = g()
    torch.cuda.synchronize()
    rh, pg = model.roi_heads, model.proposal_generator
    print("  logs", {k: [round(x, 3) for x in v.flatten().float().tolist()][:12] if torch.is_tensor(v) else str(type(v)) for k, v in rh.pending_logs.items()},
          {k: [round(x, 3) for x in v[0].flatten().tolist()] for k, v in pg.pending_logs.items()}, flush=True)
    print("replay", it, {k: round(float(v.detach()), 4) for k, v in l.items()}, "gnorm", float(opt.flat_grad.norm()), flush=True)
eager()
