"""`PeriodicCheckpointerOnlyOne` (reference cubercnn/solver/checkpoint.py:5-27): keeps only
`model_recent` every `period` iterations and `model_final` at the end.  Checkpoint I/O itself is
outside the MI355X hot path (torch.save of the state dict)."""
import os

import torch


class PeriodicCheckpointerOnlyOne:
    def __init__(self, checkpointer, period, max_iter=None, **kwargs):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        additional_state = {"iteration": iteration}
        additional_state.update(kwargs)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save("model_recent", **additional_state)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save("model_final", **additional_state)
