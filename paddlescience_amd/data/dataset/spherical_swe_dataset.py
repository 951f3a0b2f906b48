"""ppsci.data.dataset.SphericalSWEDataset (/root/reference/ppsci/data/dataset/spherical_swe_dataset.py:14-104): the shallow-water
fields on the sphere that examples/neuraloperator/train_sfno.py trains its SFNO on -- `train_SWE_<res>.npy` / `test_SWE_<res>.npy`, each a
pickled dict {"x": [n, 3, nlat, nlon], "y": [n, 3, nlat, nlon]}; no encoding, no positional channels."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional, Sequence, Tuple

import numpy as np


class SphericalSWEDataset:
    """Constructor arguments, split rule and item layout of the reference: `data_split` "train" serves the training file,
    "test_32x64" the FIRST entry of `test_resolutions`, anything else the SECOND (spherical_swe_dataset.py:80-101).  Only the served
    file is read (the reference reads all three every time)."""

    batch_index: bool = True

    def __init__(self, input_keys: Tuple[str, ...], label_keys: Tuple[str, ...], data_dir: str,
                 weight_dict: Optional[Dict[str, float]] = None, test_resolutions: Sequence[str] = ("34x64", "64x128"),
                 train_resolution: str = "34x64", data_split: str = "train"):
        self.input_keys, self.label_keys, self.data_dir = tuple(input_keys), tuple(label_keys), data_dir
        self.weight_dict = dict({k: 1.0 for k in self.label_keys}, **weight_dict) if weight_dict is not None else {}
        self.weight = self.weight_dict  # the array datasets' attribute name (Solver)
        self.test_resolutions, self.train_resolution, self.data_split = list(test_resolutions), train_resolution, data_split
        root = Path(data_dir)
        if data_split == "train":
            path = root / f"train_SWE_{train_resolution}.npy"
        elif data_split == "test_32x64":
            path = root / f"test_SWE_{self.test_resolutions[0]}.npy"
        else:
            path = root / f"test_SWE_{self.test_resolutions[1]}.npy"
        data = np.load(path.as_posix(), allow_pickle=True).item()
        self._x, self._y = np.asarray(data["x"]).astype(np.float32), np.asarray(data["y"]).astype(np.float32)

    def __len__(self):
        return self._x.shape[0]

    def __getitem__(self, index):
        return {self.input_keys[0]: self._x[index]}, {self.label_keys[0]: self._y[index]}, self.weight_dict
