"""`PeriodicCheckpointerOnlyOne` (reference cubercnn/solver/checkpoint.py:5-27, a detectron2 PeriodicCheckpointer): keeps
only `<file_prefix>_recent` every `period` iterations and `<file_prefix>_final` at the end.  Checkpoint I/O itself is
outside the MI355X hot path (torch.save of the state dict)."""


class PeriodicCheckpointerOnlyOne:
    def __init__(self, checkpointer, period, max_iter=None, max_to_keep=None, file_prefix="model"):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter
        self.max_to_keep, self.file_prefix = max_to_keep, file_prefix      # (only one periodic file is ever kept)

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        additional_state = {"iteration": iteration}
        additional_state.update(kwargs)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save("{}_recent".format(self.file_prefix), **additional_state)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save("{}_final".format(self.file_prefix), **additional_state)

    def save(self, name, **kwargs):
        self.checkpointer.save(name, **kwargs)
