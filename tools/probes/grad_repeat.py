#!/usr/bin/env python
"""Run-to-run spread of the gradient check of tests/test_model_parity.py::test_training_step_matches_reference_gpu on the GPU:
the training step of a fixture is repeated N times in its production configuration (split-K / statistics atomics included) and
the worst element-wise error of the recorded 64-element gradient heads (relative to the largest reference element, as the test
measures it) is printed per run for the worst parameters.  Usage: python tools/probes/grad_repeat.py <fixture> [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def main():
    name, runs = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import test_model_parity as T
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    from omni3d_amd.d2.events import EventStorage
    gold = torch.load(T._gold(name), weights_only=False)
    spec = gold["spec"]
    priors = synthetic.make_priors(50, bins=spec.get("prior_bins", 0))
    rows, first, spread = {}, None, {}
    for r in range(runs):
        model = MG.build_product_model(MG.product_cfg(spec["overrides"], spec.get("config", "cubercnn_DLA34_FPN.yaml")), priors, spec["seed"], device="cuda")
        batch = synthetic.make_batch(spec["images"], spec["height"], spec["width"], num_gt=spec["num_gt"], seed=spec["seed"], priors=priors)
        E = MG.variates(spec, gold["rpn_labels"].shape[1])
        model.proposal_generator.injected = {"E": E["rpn"], "proposals": gold["proposals"]}
        model.roi_heads.injected = {"E": E["roi"]}
        model.train()
        with EventStorage(0):
            losses = model(batch)
            sum(losses.values()).backward()
        grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        # run-to-run distance of the WHOLE gradient tensors (relative L2 against the first run)
        if first is None:
            first = {n: g.detach().clone() for n, g in grads.items()}
        else:
            for n, g in grads.items():
                spread.setdefault(n, []).append(float((g - first[n]).norm() / first[n].norm().clamp(min=1e-30)))
        for n, head in gold["grad_head"].items():
            g = grads[n]
            if g.dim() == 4:
                g = g.contiguous(memory_format=torch.contiguous_format)
            got = g.reshape(g.shape[0], -1).flatten()[:64].cpu() if g.dim() > 1 else g.flatten()[:64].cpu()
            rows.setdefault(n, []).append((got - head).abs().max().item() / max(head.abs().max().item(), 1e-6))
    worst = sorted(rows.items(), key=lambda kv: -max(kv[1]))[:6]
    print(name, "runs", runs)
    for n, v in worst:
        print("  %-56s" % n, " ".join("%.4f" % x for x in v))
    print("  relative L2 distance of whole gradient tensors to the first run:")
    for n, v in sorted(spread.items(), key=lambda kv: -max(kv[1]))[:5]:
        print("  %-56s" % n, " ".join("%.4f" % x for x in v))


if __name__ == "__main__":
    main()
