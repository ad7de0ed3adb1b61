"""Finds reads of uninitialised device memory: every cached allocator block is filled with a huge-int / NaN pattern before each
eager training step, and every C-ABI call is followed by a synchronize so a faulting kernel is named."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from omni3d_amd import bench_train as BT
from omni3d_amd import lib
from omni3d_amd.functional import total_loss, side_mode

side_mode("inline")
cfg, model, opt, priors = BT.build(1)
batch, packed = BT.stage_batch(model, priors, 0)

def poison():
    bufs = []
    for n, cnt in ((64, 4000), (1024, 4000), (16384, 2000), (262144, 400), (1 << 21, 200), (1 << 24, 60), (1 << 27, 24)):
        for _ in range(cnt):
            bufs.append(torch.full((n,), 0x7F7FFFFF, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    del bufs

L = lib.get()
orig = L.call
last = {"name": None}
def call(name, *a):
    last["name"] = name
    r = orig(name, *a)
    if os.environ.get("DBG_SYNC", "1") == "1":
        torch.cuda.synchronize()
    return r
L.call = call

def step():
    opt.zero_grad()
    losses = model(batch, packed)
    total = total_loss(losses)
    total.backward()
    torch.cuda.synchronize()
    return float(total), float(opt.flat_grad.abs().sum()), bool(torch.isfinite(opt.flat_grad).all())

print("clean", step(), flush=True)
for it in range(3):
    poison()
    try:
        print("poisoned", it, step(), flush=True)
    except Exception as e:
        print("exception after", last["name"], repr(e)[:300], flush=True)
        raise
