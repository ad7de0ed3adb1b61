"""detectron2.utils.comm plumbing over torch.distributed (RCCL on ROCm = backend "nccl")."""
import os
import pickle

import torch
import torch.distributed as dist


def _ok():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _ok() else 1


def get_rank():
    return dist.get_rank() if _ok() else 0


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", "0")) if _ok() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if _ok() and dist.get_world_size() > 1:
        if dist.get_backend() == dist.Backend.NCCL:
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def gather(data, dst=0, group=None):
    if get_world_size() == 1:
        return [data]
    out = [None for _ in range(get_world_size())] if get_rank() == dst else None
    dist.gather_object(data, out, dst=dst, group=group)
    return out if get_rank() == dst else []


def all_gather(data, group=None):
    if get_world_size() == 1:
        return [data]
    out = [None for _ in range(get_world_size())]
    dist.all_gather_object(out, data, group=group)
    return out


def shared_random_seed():
    ints = torch.randint(2 ** 31, (1,)).item()
    return all_gather(ints)[0]
