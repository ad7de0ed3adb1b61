"""detectron2.solver: `build_lr_scheduler` for the schedule the reference configures (configs/Base.yaml:1-9:
WarmupMultiStepLR, linear warm-up; tools/train_net.py:125 builds it, :248 steps it once per iteration).

    lr(it) = base_lr * gamma ** #{milestones <= it} * f(it),   f(it) = wf * (1 - it/W) + it/W  for it < W, else 1

(SURVEY.md Appendix A.16).  Works on any optimizer with `param_groups` (FlatSGD reads each group's "lr" at step time)."""
import bisect
import math


class WarmupMultiStepLR:
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear",
                 last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        if warmup_method not in ("linear", "constant"):
            raise ValueError("Unknown warmup method: {}".format(warmup_method))
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def _factor(self, it):
        if it >= self.warmup_iters:
            return 1.0
        if self.warmup_method == "constant":
            return self.warmup_factor
        alpha = it / self.warmup_iters
        return self.warmup_factor * (1 - alpha) + alpha

    def get_lr(self):
        k = bisect.bisect_right(self.milestones, self.last_epoch)
        return [b * self._factor(self.last_epoch) * self.gamma ** k for b in self.base_lrs]

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": self.base_lrs}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = sd["last_epoch"], list(sd["base_lrs"])
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr


class WarmupCosineLR(WarmupMultiStepLR):
    def __init__(self, optimizer, max_iters, **kw):
        self.max_iters = max_iters
        super().__init__(optimizer, [], **kw)

    def get_lr(self):
        f = self._factor(self.last_epoch)
        return [b * f * 0.5 * (1.0 + math.cos(math.pi * self.last_epoch / self.max_iters)) for b in self.base_lrs]


def build_lr_scheduler(cfg, optimizer):
    name = cfg.SOLVER.LR_SCHEDULER_NAME
    kw = dict(warmup_factor=cfg.SOLVER.WARMUP_FACTOR, warmup_iters=cfg.SOLVER.WARMUP_ITERS, warmup_method=cfg.SOLVER.WARMUP_METHOD)
    if name == "WarmupMultiStepLR":
        steps = [x for x in cfg.SOLVER.STEPS if x <= cfg.SOLVER.MAX_ITER]
        return WarmupMultiStepLR(optimizer, steps, cfg.SOLVER.GAMMA, **kw)
    if name == "WarmupCosineLR":
        return WarmupCosineLR(optimizer, cfg.SOLVER.MAX_ITER, **kw)
    raise ValueError("Unknown LR scheduler: {}".format(name))
