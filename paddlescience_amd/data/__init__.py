"""ppsci.data.build_dataloader (/root/reference/ppsci/data/__init__.py:59-209) for the array datasets.

Batches are index arrays into the dataset (no per-sample collation -- the reference also disables
auto-collation for batch-indexable datasets, :156-188).  With world_size > 1 the batch sampler becomes
rank-strided like paddle's DistributedBatchSampler (:76-99); full-batch iterable datasets refuse
world_size > 1 exactly like the reference (:62-66)."""
from __future__ import annotations

import copy
from typing import Iterator, List

import numpy as np

from ..utils import logger
from . import dataset
from .dataset import build_dataset  # noqa: F401


def _world():
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except Exception:  # noqa: BLE001
        pass
    return 1, 0


class BatchSampler:
    """Index batches; `world`/`rank` give paddle.io.DistributedBatchSampler's rank-strided sharding
    (total size padded to a multiple of world by wrapping around, /root/reference/ppsci/data/__init__.py:76-99).

    The wrapped-around samples keep every rank in lock-step (same number of batches, same batch sizes) but are
    DUPLICATES: `last_pad` tells the caller which positions of the batch just yielded are such padding -- (mask over
    this rank's batch, number of padded samples in the GLOBAL batch, size of the global batch) -- so that the loss can
    give them zero weight and normalise by the true sample count (Solver._shard_weights).  None: no
    padding in this batch."""

    def __init__(self, n: int, batch_size: int, shuffle: bool = False, drop_last: bool = False, world: int = 1,
                 rank: int = 0, seed: int = 42):
        self.n, self.batch_size, self.shuffle, self.drop_last = n, batch_size, shuffle, drop_last
        self.world, self.rank = world, rank
        self.epoch = 0
        self.rng = np.random.RandomState(seed)
        self.num_samples = int(np.ceil(n / world)) if world > 1 else n
        self.last_pad = None

    def __iter__(self) -> Iterator[np.ndarray]:
        idx = np.arange(self.n)
        if self.shuffle:
            if self.world > 1:  # every rank must draw the same permutation
                idx = np.random.RandomState(self.epoch).permutation(self.n)
                self.epoch += 1
            else:
                idx = self.rng.permutation(self.n)
        if self.world > 1:
            total = self.num_samples * self.world
            idx = np.concatenate([idx, idx[: total - self.n]])
            idx = idx[self.rank: total: self.world]
        for s in range(0, len(idx), self.batch_size):
            b = idx[s: s + self.batch_size]
            if len(b) < self.batch_size and self.drop_last:
                break
            self.last_pad = None
            if self.world > 1 and (s + len(b)) * self.world > self.n:
                # local stream position j sits at global stream position rank + j * world; positions >= n are wrap-around
                gpos = self.rank + (s + np.arange(len(b))) * self.world
                npad = (s + len(b)) * self.world - max(self.n, s * self.world)
                self.last_pad = (gpos >= self.n, int(npad), int(len(b) * self.world))
            yield b

    def __len__(self):
        if self.drop_last:
            return self.num_samples // self.batch_size
        return (self.num_samples + self.batch_size - 1) // self.batch_size


class DataLoader:
    def __init__(self, ds, batch_sampler: BatchSampler):
        self.dataset, self.batch_sampler = ds, batch_sampler

    def __iter__(self):
        for idx in self.batch_sampler:
            self.last_pad = getattr(self.batch_sampler, "last_pad", None)
            yield self.dataset[idx]

    def __len__(self):
        return len(self.batch_sampler)


class InfiniteDataLoader:
    """dataloader.py:22-47: restart the underlying loader forever."""

    def __init__(self, loader):
        self.dataloader = loader
        self.dataset = getattr(loader, "dataset", loader)

    def __iter__(self):
        while True:
            for batch in self.dataloader:
                yield batch

    def __len__(self):
        return len(self.dataloader)


def build_dataloader(_dataset, cfg):
    world, rank = _world()
    if getattr(_dataset, "is_iterable", False):
        if world > 1 and not cfg.get("shard_in_engine", False):
            # reference behaviour (data/__init__.py:62-66).  `shard_in_engine: True` is this framework's extension
            # for separable nets (BASELINE config 5): every rank draws the same tensor-product batch and the
            # SPINN engine keeps its own x-axis slab of it.
            raise ValueError(f"world_size({world}) should be 1 when using IterableDataset.")
        _dataset.shard_in_engine = bool(cfg.get("shard_in_engine", False))
        return _dataset
    cfg = copy.deepcopy({k: v for k, v in cfg.items() if k != "dataset"})
    sampler_cfg = cfg.pop("sampler", None)
    if sampler_cfg is not None:
        sampler_cfg = dict(sampler_cfg)
        name = sampler_cfg.pop("name")
        if name not in ("BatchSampler", "DistributedBatchSampler"):
            raise NotImplementedError(f"sampler {name!r}")
        bs = BatchSampler(len(_dataset), cfg["batch_size"], sampler_cfg.get("shuffle", False),
                          sampler_cfg.get("drop_last", False), world, rank, cfg.get("seed", 42))
    else:
        bs = BatchSampler(len(_dataset), cfg["batch_size"], False, False, world, rank, cfg.get("seed", 42))
        logger.message("'shuffle' and 'drop_last' are both set to False in default as sampler config is not specified.")
    loader = DataLoader(_dataset, bs)
    if len(loader) == 0:
        raise ValueError(f"batch_size({cfg['batch_size']}) should not bigger than number of samples({len(_dataset)}) "
                         f"when drop_last is {bs.drop_last}.")
    return loader
