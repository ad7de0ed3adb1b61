"""Which ingredient makes hipGraph capture from inside model(data) crash?  Each variant runs in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {
    "V1_eager_then_direct": {}, "V2_eager_then_direct_inline": {"OMNI_WGRAD_STREAM": "0"}, "V3_eager_nostep_then_direct": {},
    "V4_auto_warm0": {}, "V5_auto_warm1_inline": {"OMNI_WGRAD_STREAM": "0"}, "V7_auto_warm1": {},
    "V8_auto_warm1_bench_build": {}, "V9_eager_then_direct_fresh_batch": {},
}
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import test_autoreplay as T
    from omni3d_amd.cubercnn.solver.graphed import GraphedPipelined
    name = sys.argv[1]
    if name.startswith("V8"):
        os.environ["OMNI_BENCH_IMS"], os.environ["OMNI_BENCH_SIZE"] = "2", "128"
        from omni3d_amd import bench_train as BT, synthetic
        cfg, model, opt, priors = BT.build(1, seed=1)
        pool = [synthetic.make_batch(2, 128, 128, num_gt=4, seed=2000 + s, priors=priors) for s in range(2)]
    else:
        model, opt, pool = T._build("cuda", [], 128)

    def eager(step=True):
        ld = model(pool[0])
        losses = sum(ld.values())
        opt.zero_grad()
        losses.backward()
        if step:
            opt.step()
        torch.cuda.synchronize()

    def direct(batch):
        packed = model.prepack(batch)
        b2 = [dict(b, image=b["image"].to("cuda")) for b in batch]
        st = GraphedPipelined(model, opt, b2, packed)
        st()
        torch.cuda.synchronize()
        return len(st.stages)
    if name.startswith(("V1", "V2", "V3", "V9")):
        model.__dict__["_omni_auto"] = None
        eager(step=not name.startswith("V3"))
        print("OK", name, direct(pool[1] if name.startswith("V9") else pool[0]))
    else:
        model._omni_auto.warm = 0 if name.startswith("V4") else 1
        T._loop(model, opt, pool, 3)
        torch.cuda.synchronize()
        print("OK", name, "replays", model._omni_auto.replays, "failed", model._omni_auto.failed)
else:
    for name, env in VARIANTS.items():
        p = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        tail = (p.stdout.strip().splitlines() or [""])[-1]
        err = [ln for ln in p.stderr.splitlines() if "Error" in ln or "error" in ln][-2:]
        print(f"{name:32s} rc={p.returncode} {tail} {err}", flush=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_synthetic.py"), "--iters", "8", "--batch", "2", "--size", "128"],
                       capture_output=True, text=True, timeout=300)
    print("train_synthetic rc=%d %s" % (p.returncode, (p.stdout.strip().splitlines() or [""])[-1]), flush=True)
