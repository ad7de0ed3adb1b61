#!/usr/bin/env python
"""The non-recycled multiscale loop of the bench line alone (omni3d_amd.bench_train.dropin_loop_multiscale_stream): the AutoReplay
cache on freshly drawn batches.  usage: [OMNI_AUTO_REPLAY=0 | OMNI_AUTO_REPLAY_CACHE=n | OMNI_AUTO_REPLAY_GRIDS=...] python tools/bench_stream.py [iters]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omni3d_amd import bench_train as BT

if os.environ.get("OMNI_AUTO_REPLAY", "1") == "0":
    # all-eager reference: the loop without the cache (AutoReplay is not attached)
    import time

    import torch
    from omni3d_amd import synthetic
    cfg, model, opt, priors = BT.build(1, seed=2)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    stream = [synthetic.make_multiscale_batch(BT.IMS_PER_GPU, 7000 + s, priors=priors) for s in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = []
    for b in stream:
        t1 = time.perf_counter()
        loss = sum(model(b).values())
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t1))
    print(json.dumps({"eager_ms_per_iteration": 1e3 * (time.perf_counter() - t0) / n, "first_10": [round(t, 1) for t in ts[:10]],
                      "last_10": [round(t, 1) for t in ts[-10:]]}))
else:
    print(json.dumps(BT.dropin_loop_multiscale_stream(int(sys.argv[1]) if len(sys.argv) > 1 else 200)))
