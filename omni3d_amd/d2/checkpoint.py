"""detectron2.checkpoint.DetectionCheckpointer plumbing (fvcore Checkpointer semantics): `save(name, **extra)` writes
`<save_dir>/<name>.pth` = {"model": state_dict, <checkpointable name>: its state_dict..., **extra} and records it in
`<save_dir>/last_checkpoint`; `resume_or_load(path, resume=)` / `load(path, checkpointables=)` restore them and return the
extra data (tools/train_net.py:128-145).  Checkpoint I/O is outside the MI355X hot path; state-dict names are the
reference's (tests/test_checkpoint.py), so released .pth files of the model zoo load."""
import os

import torch


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
        self.model = model.module if hasattr(model, "module") and isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
        self.save_dir = save_dir
        self.checkpointables = dict(checkpointables)
        from . import comm
        self.save_to_disk = comm.is_main_process() if save_to_disk is None else save_to_disk

    def add_checkpointable(self, key, obj):
        self.checkpointables[key] = obj

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        for key, obj in self.checkpointables.items():
            data[key] = obj.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        basename = "{}.pth".format(name)
        torch.save(data, os.path.join(self.save_dir, basename))
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(basename)

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return os.path.join(self.save_dir, f.read().strip())
        except IOError:
            return ""

    def load(self, path, checkpointables=None):
        if not path or "://" in path and not os.path.exists(path):
            return {}                       # '' / synthetic://random-init / un-downloadable URLs: keep the initialisation
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        state = ckpt.pop("model", ckpt)
        self.model.load_state_dict(state, strict=False)
        for key in (self.checkpointables if checkpointables is None else checkpointables):
            if key in ckpt:
                self.checkpointables[key].load_state_dict(ckpt.pop(key))
        return ckpt

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])
