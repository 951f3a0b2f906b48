"""Array datasets (/root/reference/ppsci/data/dataset/array_dataset.py:29-231): the collocation clouds of
a constraint as named [n,1] numpy arrays."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

from ...utils import logger


class NamedArrayDataset:
    """Batch-indexable dataset (array_dataset.py:29-85)."""

    batch_index: bool = True

    def __init__(self, input: Dict[str, np.ndarray], label: Optional[Dict[str, np.ndarray]] = None,
                 weight: Optional[Dict[str, np.ndarray]] = None, transforms=None):
        self.input = input
        self.label = {} if label is None else label
        self.input_keys = tuple(input.keys())
        self.label_keys = tuple(self.label.keys())
        self.weight = {} if weight is None else weight
        self.transforms = transforms
        self._len = len(next(iter(input.values())))
        for key in input:
            if key in self.label and len(input[key]) != len(self.label[key]):
                logger.warning(f"The length of input {key}({len(input[key])}) is not equal to the length of label "
                               f"{key}({len(self.label[key])}).")

    def __getitem__(self, idx):
        item = ({k: v[idx] for k, v in self.input.items()}, {k: v[idx] for k, v in self.label.items()},
                {k: v[idx] for k, v in self.weight.items()})
        if self.transforms is not None:
            item = self.transforms(*item)
        return item

    def __len__(self):
        return self._len


class IterableNamedArrayDataset:
    """One full batch, identical every iteration (array_dataset.py:88-151); world_size must be 1."""

    batch_index: bool = False
    is_iterable = True

    def __init__(self, input: Dict[str, np.ndarray], label: Optional[Dict[str, np.ndarray]] = None,
                 weight: Optional[Dict[str, np.ndarray]] = None, transforms=None):
        self.input = {k: np.asarray(v) for k, v in input.items()}
        self.label = {k: np.asarray(v) for k, v in label.items()} if label is not None else {}
        self.input_keys = tuple(input.keys())
        self.label_keys = tuple(self.label.keys())
        self.weight = {k: np.asarray(v, dtype="float32") for k, v in weight.items()} if weight is not None else None
        self._len = len(next(iter(self.input.values())))
        self.transforms = transforms

    @property
    def num_samples(self):
        return self._len

    def __iter__(self):
        if callable(self.transforms):
            yield self.transforms(self.input, self.label, self.weight)
        else:
            yield self.input, self.label, self.weight

    def __len__(self):
        return 1


class ContinuousNamedArrayDataset:
    """Fresh samples every iteration from user callables (array_dataset.py:154-231)."""

    batch_index: bool = False
    is_iterable = True

    def __init__(self, input: Callable, label: Callable, weight: Optional[Callable] = None, transforms=None):
        self.input_fn, self.input_keys = input, tuple(self_keys(input()))
        self.label_fn = label
        ref_in = self.input_fn()
        self.label_keys = tuple(self.label_fn(ref_in).keys())
        self.weight_fn = weight
        self.transforms = transforms

    @property
    def num_samples(self):
        raise NotImplementedError("ContinuousNamedArrayDataset has no fixed number of samples.")

    def __iter__(self):
        def to_f32(d):
            return {k: np.asarray(v, dtype="float32") for k, v in d.items()}

        while True:
            inp = self.input_fn()
            lab = self.label_fn(inp)
            w = self.weight_fn(inp) if callable(self.weight_fn) else None
            item = (to_f32(inp), to_f32(lab), to_f32(w) if w is not None else None)
            if callable(self.transforms):
                item = self.transforms(*item)
            yield item

    def __len__(self):
        return 1


def self_keys(d):
    return d.keys()
