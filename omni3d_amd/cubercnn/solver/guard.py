"""The training loop's safety logic of the reference (tools/train_net.py:157-285, 471-498) as device-side state plus ONE
small all-reduce per step.

Reference per iteration: `allreduce_dict(loss_dict)` + ten `.item()` (:186), `comm.synchronize()` (:190), the rolling-loss
divergence test (:198-215), the per-parameter NaN/Inf scan (:222-233), an all-reduce of the "diverging" flag (:237-243),
skip or step (:245-253), the retry decision and an all-reduce of it (:258-270) -- three collectives, three barriers and a
dozen host syncs.  Here:

  * one 12-float vector per rank [10 losses | sum | non-finite-gradient flag] is all-reduced ONCE (sum);
  * the divergence test, the rolling mean, the success / explode counters and the retry decision are evaluated from the
    reduced vector by one one-wave kernel (csrc/optim.hip guard_post_kernel), identically on every rank (same inputs => same decision, so the flags
    need no second collective), and `skip` lands in the device float the fused SGD kernel reads (`FlatSGD.skip_flag`);
  * the host reads (skipped, retry, 10 reduced losses) back in one copy -- every step (`sync=True`, the reference's
    semantics: it logs the scalars and may return False to restart) or only when it wants to look (`sync=False`).

The clipped-loss trick of the reference (`losses.clip(0, 1)` before backward when diverging, then zero_grad and no step)
only exists to keep that iteration's backward finite; its net effect -- no update -- is what the skip flag does."""
import torch
import torch.distributed as dist

TOLERANCE = 4.0      # tools/train_net.py:163
GAMMA = 0.02         # :165


class StepGuard:
    def __init__(self, loss_names, stabilize, checkpoint_period, device, group=None):
        self.names = sorted(loss_names)                      # allreduce_dict sorts the keys (:486)
        self.stabilize = float(stabilize)                    # cfg.MODEL.STABILIZE
        self.half_period = 0.5 * float(checkpoint_period)    # :259
        self.group = group
        n = len(self.names)
        self.vec = torch.zeros(n + 2, dtype=torch.float32, device=device)       # [losses | total | nonfinite]
        self.state = torch.tensor([float("nan"), 0.0, 0.0], dtype=torch.float32, device=device)   # [recent (NaN = None, :166), success, explode]
        self.skip = torch.zeros(1, dtype=torch.float32, device=device)          # -> FlatSGD.skip_flag
        self.out = torch.zeros(n + 3, dtype=torch.float32, device=device)       # [skipped, retry, total, losses...]

    @property
    def nonfinite_flag(self):
        """(1,) view the fused gradient scan writes into (FlatSGD.check_nonfinite)"""
        return self.vec[-1:]

    def world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    @torch.no_grad()
    def update(self, loss_dict, sync=True):
        """Call after backward + all-reduce of the gradients + check_nonfinite(self.nonfinite_flag), before optimizer.step().
        -> (skipped, retry, {name: reduced loss}) as host values when sync, else None."""
        from ...kernels import det
        n = len(self.names)
        from ...kernels import glue
        vals = [loss_dict[k].detach() for k in self.names]
        if not glue.guard_gather(vals, self.vec):                    # vec[:n] = the losses, vec[n] = their sum: one launch (round 6)
            torch.stack([v.float().reshape(()) for v in vals], out=self.vec[:n])
            det.guard_pre(self.vec, n)                               # vec[n] = sum of the losses
        w = self.world()
        if w > 1:
            dist.all_reduce(self.vec, group=self.group)              # the ONE collective of the guard
        # average, divergence test (:194-215), counters, retry (:258-259), skip flag, host record -- one one-wave kernel
        det.guard_post(self.vec, n, w, self.stabilize, self.half_period, TOLERANCE, GAMMA, self.state, self.skip, self.out)
        if not sync:
            return None
        return self.read()

    def read(self):
        v = self.out.tolist()                                                     # one device->host copy
        return bool(v[0]), bool(v[1]), dict(zip(self.names, v[3:]), total_loss=v[2])


def allreduce_dict(input_dict, average=True, group=None):
    """tools/train_net.py:471-498: every value a 0-d tensor; one stacked all-reduce in sorted-key order."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values, group=group)
        if average:
            values /= world
        return {k: v for k, v in zip(names, values)}
