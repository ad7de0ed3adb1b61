"""detectron2.engine plumbing used by tools/train_net.py:15-20: argument parser, setup, writers, `launch`."""
import argparse
import json
import os
import sys

import torch


def default_argument_parser(epilog=None):
    p = argparse.ArgumentParser(epilog=epilog, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    port = 2 ** 15 + 2 ** 14 + hash(os.getuid() if sys.platform != "win32" else 1) % 2 ** 14
    p.add_argument("--dist-url", default="tcp://127.0.0.1:{}".format(port))
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def default_setup(cfg, args):
    from . import comm
    out = cfg.OUTPUT_DIR
    if comm.is_main_process() and out:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "config.yaml"), "w") as f:
            f.write(cfg.dump() if hasattr(cfg, "dump") else str(cfg))
    seed = getattr(cfg, "SEED", -1)
    if seed is not None and seed >= 0:
        torch.manual_seed(seed + comm.get_rank())


class JSONWriter:
    """metrics.json lines of EventStorage scalars (detectron2 JSONWriter, reduced to the latest values)"""

    def __init__(self, json_file):
        self._fh = open(json_file, "a")

    def write(self):
        from .events import get_event_storage
        st = get_event_storage()
        rec = {"iteration": st.iter}
        rec.update({k: v[0] for k, v in st.latest().items()})
        self._fh.write(json.dumps(rec, sort_keys=True) + "\n")
        self._fh.flush()

    def close(self):
        self._fh.close()


def default_writers(output_dir, max_iter=None):
    if not output_dir:
        return []
    os.makedirs(output_dir, exist_ok=True)
    return [JSONWriter(os.path.join(output_dir, "metrics.json"))]


def _worker(local_rank, main_func, world, num_gpus_per_machine, machine_rank, dist_url, args):
    import torch.distributed as dist
    rank = machine_rank * num_gpus_per_machine + local_rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["LOCAL_RANK"] = str(local_rank)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", init_method=dist_url, world_size=world, rank=rank, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo", init_method=dist_url, world_size=world, rank=rank)
    try:
        main_func(*args)
    finally:
        dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=(), timeout=None):
    """one process per GPU over RCCL (backend "nccl" on ROCm); a single process runs in place"""
    world = num_machines * num_gpus_per_machine
    if world <= 1:
        return main_func(*args)
    import torch.multiprocessing as mp
    mp.spawn(_worker, nprocs=num_gpus_per_machine, args=(main_func, world, num_gpus_per_machine, machine_rank, dist_url, args), daemon=False)
