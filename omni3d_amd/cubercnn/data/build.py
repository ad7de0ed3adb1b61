"""`build_detection_train_loader` / `build_detection_test_loader` (reference cubercnn/data/build.py:44-231).

The reference's loader is torch DataLoader workers around JPEG decoding -- host-side I/O, outside the MI355X hot path
(SURVEY.md 8b).  What the training loop needs from it is kept: an endless iterator of `IMS_PER_BATCH / world` mapped dicts
per step, drawn by a seeded shuffling sampler sharded over ranks (detectron2 TrainingSampler), from the dicts registered in
`DatasetCatalog` under cfg.DATASETS.TRAIN, with the reference's category repeat factors (RepeatFactorTrainingSampler) and
per-source balancing (DATALOADER.BALANCE_DATASETS, build.py:60-121)."""
import itertools

import numpy as np
import torch

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


def repeat_factors_from_category_frequency(dataset_dicts, repeat_thresh):
    """build.py:127-172 (LVIS repeat-factor sampling): r(I) = max_{c in I} max(1, sqrt(t / f(c)))"""
    import math
    from collections import defaultdict
    freq = defaultdict(int)
    for d in dataset_dicts:
        for c in {a["category_id"] for a in d["annotations"]}:
            if c >= 0:
                freq[c] += 1
    n = len(dataset_dicts)
    rep = {c: max(1.0, math.sqrt(repeat_thresh / (v / n))) for c, v in freq.items()}
    out = [max({rep[c] for c in {a["category_id"] for a in d["annotations"]} if c >= 0}, default=1.0) for d in dataset_dicts]
    return torch.tensor(out, dtype=torch.float32)


def dataset_balance_weights(dataset_dicts, dataset_id_to_src):
    """build.py:66-91 (DATALOADER.BALANCE_DATASETS): images of a source that contributes the fraction p of the training set get
    the weight (1 - p) / min over sources of (1 - p); a single source gives all ones"""
    assert dataset_id_to_src is not None, "Need dataset sources."
    sources = sorted(set(dataset_id_to_src.values()), key=str)
    src_of = np.array([sources.index(dataset_id_to_src[d["dataset_id"]]) for d in dataset_dicts])
    present = np.unique(src_of)
    if len(present) == 1:
        return torch.ones(len(src_of), dtype=torch.float32)
    counts = np.bincount(src_of, minlength=len(sources))[present].astype(np.float64)
    w = 1.0 - counts / counts.sum()
    w = w / w.min()
    out = torch.zeros(len(src_of), dtype=torch.float32)
    for s_id, wt in zip(present, w):
        out[torch.from_numpy(src_of == s_id)] = float(wt)
    return out


class _TrainLoader:
    """endless batches; index stream = detectron2 TrainingSampler (shuffled epochs) or RepeatFactorTrainingSampler (each epoch
    repeats image i floor(r_i) times plus once more with probability frac(r_i), then shuffles), one shared seeded stream
    sharded over ranks by position"""

    def __init__(self, dataset, mapper, batch, seed=0, repeat_factors=None):
        self.dataset, self.mapper, self.batch = dataset, mapper, batch
        self.seed, self.rank, self.world = seed, comm.get_rank(), comm.get_world_size()
        self.repeat_factors = repeat_factors

    def _epoch(self, g):
        if self.repeat_factors is None:
            return torch.randperm(len(self.dataset), generator=g).tolist()
        ip, fp = torch.trunc(self.repeat_factors), self.repeat_factors - torch.trunc(self.repeat_factors)
        reps = ip + (torch.rand(len(fp), generator=g) < fp).float()
        idx = torch.repeat_interleave(torch.arange(len(reps)), reps.long())
        return idx[torch.randperm(len(idx), generator=g)].tolist()

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed)

        def stream():
            while True:
                yield from self._epoch(g)
        mine = itertools.islice(stream(), self.rank, None, self.world)
        while True:
            yield [self.mapper(self.dataset[i]) for i in itertools.islice(mine, self.batch)]


def build_detection_train_loader(cfg, mapper=None, *, dataset=None, sampler=None, dataset_id_to_src=None, total_batch_size=None,
                                 aspect_ratio_grouping=None, num_workers=0):
    if dataset is None:
        dataset = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    if mapper is None:
        from .dataset_mapper import DatasetMapper3D
        mapper = DatasetMapper3D(cfg, True)
    name = cfg.DATALOADER.SAMPLER_TRAIN
    if name not in ("TrainingSampler", "RepeatFactorTrainingSampler"):
        raise ValueError("Unknown training sampler: {}".format(name))
    rf = repeat_factors_from_category_frequency(dataset, cfg.DATALOADER.REPEAT_THRESHOLD) if name == "RepeatFactorTrainingSampler" else None
    if getattr(cfg.DATALOADER, "BALANCE_DATASETS", False):
        w = dataset_balance_weights(dataset, dataset_id_to_src)
        if rf is None:                   # TrainingSampler + balancing = repeat factors equal to the source weights (build.py:104-105)
            rf = w
        else:                            # categories AND sources (build.py:115-121)
            rf = rf * w
            rf = rf / rf.min().item()
    total = cfg.SOLVER.IMS_PER_BATCH if total_batch_size is None else total_batch_size
    world = comm.get_world_size()
    assert total > 0 and total % world == 0, "Total batch size ({}) must be divisible by the number of gpus ({}).".format(total, world)
    seed = int(getattr(cfg, "SEED", -1))
    if seed < 0:          # detectron2 TrainingSampler: no configured seed = one random seed shared by all ranks
        seed = comm.shared_random_seed()
    return _TrainLoader(dataset, mapper, total // world, seed=seed, repeat_factors=rf)


class _TestLoader:
    """Lazy batches over this rank's shard: a batch is decoded / resized when the evaluation loop asks for it (optionally one
    batch of look-ahead in a helper thread; off by default because the mapper's resize launches device work on the caller's
    stream), never the whole split up front -- Omni3D's test splits are tens of thousands of images.
    Shards are detectron2 InferenceSampler's: contiguous ranges, the first `len % world` ranks one image longer, so the
    gathered predictions keep the dataset order."""

    def __init__(self, dataset, mapper, batch_size, rank, world, prefetch=False):
        self.dataset, self.mapper, self.batch_size, self.prefetch = dataset, mapper, max(int(batch_size), 1), prefetch
        n = len(dataset)
        size, extra = n // world, n % world
        self.begin = size * rank + min(rank, extra)
        self.end = self.begin + size + (1 if rank < extra else 0)

    def __len__(self):
        return (self.end - self.begin + self.batch_size - 1) // self.batch_size

    def _batch(self, k):
        lo = self.begin + k * self.batch_size
        return [self.mapper(self.dataset[i]) for i in range(lo, min(lo + self.batch_size, self.end))]

    def __getitem__(self, k):
        if not 0 <= k < len(self):
            raise IndexError(k)
        return self._batch(k)

    def __iter__(self):
        n = len(self)
        if not self.prefetch or n <= 1:
            for k in range(n):
                yield self._batch(k)
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(self._batch, 0)
            for k in range(n):
                cur = nxt.result()
                if k + 1 < n:
                    nxt = pool.submit(self._batch, k + 1)
                yield cur


def build_detection_test_loader(cfg=None, dataset_name=None, mapper=None, *, dataset=None, batch_size=1, num_workers=0):
    if dataset is None:
        dataset = get_detection_dataset_dicts(dataset_name, filter_empty=False)
    if mapper is None:
        from .dataset_mapper import DatasetMapper3D
        mapper = DatasetMapper3D(cfg, False)
    return _TestLoader(dataset, mapper, batch_size, comm.get_rank(), comm.get_world_size())
