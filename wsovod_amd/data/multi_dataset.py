"""Host side of the mixed-dataset path: which image each rank draws next and how draws are
grouped into single-dataset batches.  Index work only -- the stream is bit-identical to the reference's for
the same seed (tests/golden/g11_multi_dataset_sampler.npz).

Mirrors /root/reference/wsovod/data/samplers/distributed_sampler_multi_dataset.py:17-137
(`MultiDatasetTrainingSampler`) and /root/reference/wsovod/data/build_multi_dataset.py:540-577
(`MultiDatasetAspectRatioGroupedDataset`).
"""
import itertools
import math
from collections import Counter

import torch
import torch.distributed as dist

__all__ = ["MultiDatasetTrainingSampler", "MultiDatasetAspectRatioGroupedDataset", "repeat_factors_from_category_frequency"]


def _categories(dataset_dict):
    return {ann["category_id"] for ann in dataset_dict["annotations"]}


def repeat_factors_from_category_frequency(dataset_dicts, repeat_thresh):
    """LVIS repeat-factor sampling (detectron2 RepeatFactorTrainingSampler, un-vendored):
    f(c) = fraction of images containing c, r(c) = max(1, sqrt(t / f(c))), r(image) = max over its categories."""
    freq = Counter()
    for d in dataset_dicts:
        freq.update(_categories(d))
    n = len(dataset_dicts)
    cat_rep = {c: max(1.0, math.sqrt(repeat_thresh / (v / n))) for c, v in freq.items()}
    return torch.tensor([max({cat_rep[c] for c in _categories(d)}, default=1.0) for d in dataset_dicts],
                        dtype=torch.float32)


class MultiDatasetTrainingSampler(torch.utils.data.Sampler):
    """Infinite stream of dataset indices; every image is repeated `repeat_factor` times per epoch (the
    fractional part by stochastic rounding), epochs are shuffled, and rank r takes every world_size-th draw
    starting at r.  All ranks run the same generator, so the shards are disjoint by construction."""

    def __init__(self, repeat_factors, *, shuffle=True, seed=None, rank=None, world_size=None):
        self._shuffle = shuffle
        if seed is None:
            raise ValueError("seed must be shared by all ranks (the reference draws it with comm.shared_random_seed)")
        self._seed = int(seed)
        ddp = dist.is_available() and dist.is_initialized()
        self._rank = rank if rank is not None else (dist.get_rank() if ddp else 0)
        self._world_size = world_size if world_size is not None else (dist.get_world_size() if ddp else 1)
        repeat_factors = torch.as_tensor(repeat_factors, dtype=torch.float32)
        self._int_part = torch.trunc(repeat_factors)
        self._frac_part = repeat_factors - self._int_part

    @staticmethod
    def get_repeat_factors(dataset_dicts, num_datasets, dataset_ratio, use_rfs, use_cas, repeat_thresh, cas_lambda):
        """Per-image factor = (largest dataset size / this dataset's size) * ratio * per-image balance term
        (1, LVIS repeat factor, or class-aware factor normalised to mean 1).  `dataset_dicts` is the
        concatenation of the datasets in id order."""
        sizes = [0] * num_datasets
        for d in dataset_dicts:
            sizes[d["dataset_id"]] += 1
        assert len(dataset_ratio) == len(sizes), (len(dataset_ratio), len(sizes))
        largest = max(sizes)
        factors, start = [], 0
        for i, s in enumerate(sizes):
            assert not (use_rfs[i] and use_cas[i])
            part = dataset_dicts[start:start + s]
            if use_rfs[i]:
                f = repeat_factors_from_category_frequency(part, repeat_thresh)
            elif use_cas[i]:
                f = MultiDatasetTrainingSampler.get_class_balance_factor_per_dataset(part, l=cas_lambda)
                f = f * (s / f.sum())
            else:
                f = torch.ones(s, dtype=torch.float32)
            weight = torch.ones(s, dtype=torch.float32) * largest / s * dataset_ratio[i]
            factors.append(weight * f)
            start += s
        return torch.cat(factors)

    @staticmethod
    def get_class_balance_factor_per_dataset(dataset_dicts, l=1.0):
        freq = Counter()
        for d in dataset_dicts:
            freq.update(_categories(d))
        return torch.tensor([sum(1.0 / (freq[c] ** l) for c in _categories(d)) for d in dataset_dicts],
                            dtype=torch.float32)

    def _get_epoch_indices(self, generator):
        rands = torch.rand(len(self._frac_part), generator=generator)
        reps = (self._int_part + (rands < self._frac_part).float()).to(torch.int64)
        return torch.repeat_interleave(torch.arange(len(reps), dtype=torch.int64), reps)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            indices = self._get_epoch_indices(g)
            if self._shuffle:
                indices = indices[torch.randperm(len(indices), generator=g)]
            yield from indices.tolist()

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)


class MultiDatasetAspectRatioGroupedDataset(torch.utils.data.IterableDataset):
    """Group a stream of mapped dicts into batches that hold ONE dataset and one orientation (w > h or not):
    the model reads `batched_inputs[0]["dataset_id"]` for the whole batch, and like-shaped images pad less.
    `batch_size[dataset_id]` images per batch."""

    def __init__(self, dataset, batch_size, num_datasets):
        self.dataset = dataset
        self.batch_size = batch_size
        self._buckets = [[] for _ in range(2 * num_datasets)]

    def __iter__(self):
        for d in self.dataset:
            key = 2 * d["dataset_id"] + (0 if d["width"] > d["height"] else 1)
            bucket = self._buckets[key]
            bucket.append(d)
            if len(bucket) == self.batch_size[d["dataset_id"]]:
                self._buckets[key] = []
                yield bucket
