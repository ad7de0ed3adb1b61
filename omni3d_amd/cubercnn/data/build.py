"""`build_detection_train_loader` / `build_detection_test_loader` (reference cubercnn/data/build.py:44-231).

The reference's loader is torch DataLoader workers around JPEG decoding -- host-side I/O, outside the MI355X hot path
(SURVEY.md 8b).  What the training loop needs from it is kept: an endless iterator of `IMS_PER_BATCH / world` mapped dicts
per step, drawn by a seeded shuffling sampler sharded over ranks (detectron2 TrainingSampler), from the dicts registered in
`DatasetCatalog` under cfg.DATASETS.TRAIN.  Dataset balancing / repeat-factor sampling (build.py:60-112) are not built."""
import itertools

import numpy as np

from ...d2 import comm
from ...d2.data import DatasetCatalog


def get_detection_dataset_dicts(names, filter_empty=True, **kwargs):
    if isinstance(names, str):
        names = [names]
    assert len(names), names
    dicts = [DatasetCatalog.get(n) for n in names]
    for n, d in zip(names, dicts):
        assert len(d), "Dataset '{}' is empty!".format(n)
    dicts = list(itertools.chain.from_iterable(dicts))
    if filter_empty and "annotations" in dicts[0]:
        dicts = [d for d in dicts if any(a.get("iscrowd", 0) == 0 for a in d["annotations"])]
    assert len(dicts), "No valid data found in {}.".format(",".join(names))
    return dicts


class _TrainLoader:
    def __init__(self, dataset, mapper, batch, seed=0):
        self.dataset, self.mapper, self.batch = dataset, mapper, batch
        self.seed, self.rank, self.world = seed, comm.get_rank(), comm.get_world_size()

    def __iter__(self):
        rs = np.random.RandomState(self.seed)
        def stream():
            while True:
                yield from rs.permutation(len(self.dataset)).tolist()
        mine = itertools.islice(stream(), self.rank, None, self.world)          # TrainingSampler: one shared permutation stream
        while True:
            yield [self.mapper(self.dataset[i]) for i in itertools.islice(mine, self.batch)]


def build_detection_train_loader(cfg, mapper=None, *, dataset=None, sampler=None, dataset_id_to_src=None, total_batch_size=None,
                                 aspect_ratio_grouping=None, num_workers=0):
    if dataset is None:
        dataset = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    if mapper is None:
        from .dataset_mapper import DatasetMapper3D
        mapper = DatasetMapper3D(cfg, True)
    if cfg.DATALOADER.SAMPLER_TRAIN != "TrainingSampler" or getattr(cfg.DATALOADER, "BALANCE_DATASETS", False):
        raise NotImplementedError("MI355X hot path: TrainingSampler without dataset balancing (host-side sampling variants are out of scope)")
    total = cfg.SOLVER.IMS_PER_BATCH if total_batch_size is None else total_batch_size
    world = comm.get_world_size()
    assert total > 0 and total % world == 0, "Total batch size ({}) must be divisible by the number of gpus ({}).".format(total, world)
    return _TrainLoader(dataset, mapper, total // world, seed=int(getattr(cfg, "SEED", 0) if getattr(cfg, "SEED", -1) >= 0 else 0))


class _TestLoader(list):
    pass


def build_detection_test_loader(cfg=None, dataset_name=None, mapper=None, *, dataset=None, batch_size=1, num_workers=0):
    if dataset is None:
        dataset = get_detection_dataset_dicts(dataset_name, filter_empty=False)
    if mapper is None:
        from .dataset_mapper import DatasetMapper3D
        mapper = DatasetMapper3D(cfg, False)
    rank, world = comm.get_rank(), comm.get_world_size()
    shard = dataset[rank::world]                                                  # InferenceSampler shards contiguous-ish ranges
    out = _TestLoader([[mapper(d) for d in shard[i:i + batch_size]] for i in range(0, len(shard), batch_size)])
    out.dataset = dataset
    return out
