"""ppsci.data.dataset.DarcyFlowDataset with its helpers UnitGaussianNormalizer / PositionalEmbedding2D
(/root/reference/ppsci/data/dataset/darcyflow_dataset.py:25-296): the `.npy` dict format (`x` = permeability
[n, H, W], `y` = pressure [n, H, W]) of `darcy_train_<res>.npy` / `darcy_test_<res>.npy`, channel dimension inserted
at `channel_dim`, optional unit-Gaussian encoding of inputs / outputs (statistics of the TRAINING split, `encode_output`
applied to the training labels only -- the validators compare in physical units), and the positional encoding: the
x- and y-coordinates `linspace(lo, hi, n + 1)[:-1]` appended as two channels.

Host-side numpy (the reference builds paddle CPU tensors in __init__ / __getitem__).  Batch-indexable like the array
datasets: `ds[idx_array]` returns the stacked batch, so the framework's BatchSampler path needs no collate function."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional, Sequence, Tuple

import numpy as np


class UnitGaussianNormalizer:
    """darcyflow_dataset.py:25-69: x -> (x - mean) / (std + eps); std is the UNBIASED estimate (paddle.std)."""

    def __init__(self, x: np.ndarray, eps: float = 1e-7, reduce_dim: Sequence[int] = (0,), verbose: bool = False):
        n_samples, *shape = x.shape
        self.sample_shape, self.verbose, self.reduce_dim, self.eps = shape, verbose, list(reduce_dim), eps
        ax = tuple(self.reduce_dim)
        self.mean = np.mean(x, axis=ax, keepdims=True, dtype=np.float64).astype(x.dtype).squeeze(0)
        self.std = np.std(x, axis=ax, keepdims=True, ddof=1, dtype=np.float64).astype(x.dtype).squeeze(0)
        if verbose:
            print(f"UnitGaussianNormalizer init on {n_samples}, reducing over {self.reduce_dim}, samples of shape {shape}.")
            print(f"   Mean and std of shape {self.mean.shape}, eps={eps}")

    def encode(self, x: np.ndarray) -> np.ndarray:
        return (x - self.mean) / (self.std + self.eps)

    def decode(self, x, sample_idx=None):
        if sample_idx is None:
            std, mean = self.std + self.eps, self.mean
        elif self.mean.ndim == np.ndim(sample_idx[0]):
            std, mean = self.std[sample_idx] + self.eps, self.mean[sample_idx]
        else:
            std, mean = self.std[:, sample_idx] + self.eps, self.mean[:, sample_idx]
        return x * std + mean


def regular_grid(spatial_dims: Tuple[int, int], grid_boundaries=((0, 1), (0, 1))) -> Tuple[np.ndarray, np.ndarray]:
    """darcyflow_dataset.py:105-120: cell-corner coordinates, `ij` indexing."""
    height, width = spatial_dims
    xt = np.linspace(grid_boundaries[0][0], grid_boundaries[0][1], height + 1, dtype=np.float32)[:-1]
    yt = np.linspace(grid_boundaries[1][0], grid_boundaries[1][1], width + 1, dtype=np.float32)[:-1]
    return np.meshgrid(xt, yt, indexing="ij")


def get_grid_positional_encoding(input_tensor, grid_boundaries=((0, 1), (0, 1)), channel_dim: int = 1):
    """darcyflow_dataset.py:72-102."""
    shape = list(np.shape(input_tensor))
    height, width = shape[-2:]
    gx, gy = regular_grid((height, width), grid_boundaries)
    if len(shape) == 2:
        return np.expand_dims(gx, channel_dim), np.expand_dims(gy, channel_dim)
    return np.expand_dims(gx[None], channel_dim), np.expand_dims(gy[None], channel_dim)


class PositionalEmbedding2D:
    """darcyflow_dataset.py:123-165: appends the x / y grids as channels; caches the grid of the last resolution."""

    def __init__(self, grid_boundaries=((0, 1), (0, 1))):
        self.grid_boundaries = grid_boundaries
        self._grid = None
        self._res = None

    def grid(self, spatial_dims, dtype):
        spatial_dims = tuple(int(s) for s in spatial_dims)
        if self._grid is None or self._res != spatial_dims:
            gx, gy = regular_grid(spatial_dims, self.grid_boundaries)
            self._grid = gx.astype(dtype)[None, None], gy.astype(dtype)[None, None]
            self._res = spatial_dims
        return self._grid

    def __call__(self, data: np.ndarray) -> np.ndarray:
        """[C, H, W] -> [C + 2, H, W]; batched [B, C, H, W] -> [B, C + 2, H, W] (this framework's batch-index path)."""
        single = data.ndim == 3
        if single:
            data = data[None]
        x, y = self.grid(data.shape[-2:], data.dtype)
        n = data.shape[0]
        out = np.concatenate((data, np.broadcast_to(x, (n,) + x.shape[1:]), np.broadcast_to(y, (n,) + y.shape[1:])), axis=1)
        return out[0] if single else out


class DarcyFlowDataset:
    """ppsci.data.dataset.DarcyFlowDataset (darcyflow_dataset.py:168-296): constructor arguments, item layout and the
    `input_encoder` / `output_encoder` attributes of the reference; organised around ONE served split.

    `data_split` picks what `__getitem__` serves: "train" (`darcy_train_<train_resolution>.npy`), "test_16x16" (the FIRST
    entry of `test_resolutions`) or anything else (the SECOND) -- the reference's rule.  The normalisation statistics
    always come from the training file; the output encoding is applied to the training labels only (validators compare in
    physical units).  Only the training file and the served split are read (the reference reads all three every time)."""

    batch_index: bool = True
    _SPLIT_OF = {"train": None, "test_16x16": 0}  # anything else: the second test resolution

    def __init__(self, input_keys: Tuple[str, ...], label_keys: Tuple[str, ...], data_dir: str,
                 weight_dict: Optional[Dict[str, float]] = None, test_resolutions: Sequence[int] = (32,),
                 train_resolution: int = 32, grid_boundaries=((0, 1), (0, 1)), positional_encoding: bool = True,
                 encode_input: bool = False, encode_output: bool = True, encoding: str = "channel-wise",
                 channel_dim: int = 1, data_split: str = "train"):
        bad = [r for r in test_resolutions if r not in (16, 32)]
        if bad:
            raise ValueError(f"test resolutions must be 16 or 32 (the published Darcy test sets), but got {list(test_resolutions)}")
        if encoding not in ("channel-wise", "pixel-wise"):
            raise ValueError(f"encoding={encoding!r}")
        self.input_keys, self.label_keys, self.data_dir = tuple(input_keys), tuple(label_keys), data_dir
        self.weight_dict = dict({k: 1.0 for k in self.label_keys}, **weight_dict) if weight_dict is not None else {}
        self.weight = self.weight_dict  # the array datasets' attribute name (Solver)
        self.test_resolutions, self.train_resolution = list(test_resolutions), train_resolution
        self.grid_boundaries, self.positional_encoding = grid_boundaries, positional_encoding
        self.encode_input, self.encode_output, self.encoding = encode_input, encode_output, encoding
        self.channel_dim, self.data_split = channel_dim, data_split

        root = Path(data_dir)
        x_train, y_train = self._read(root / f"darcy_train_{train_resolution}.npy")
        which = self._SPLIT_OF.get(data_split, 1)
        if which is None:
            xs, ys = x_train, y_train
        else:
            xs, ys = self._read(root / f"darcy_test_{self.test_resolutions[which]}.npy")
        # statistics of the TRAINING split; "channel-wise": one mean / std over everything, "pixel-wise": per pixel
        axes = list(range(x_train.ndim)) if encoding == "channel-wise" else [0]
        self.input_encoder = UnitGaussianNormalizer(x_train, reduce_dim=axes) if encode_input else None
        self.output_encoder = UnitGaussianNormalizer(y_train, reduce_dim=axes) if encode_output else None
        if self.input_encoder is not None:
            xs = self.input_encoder.encode(xs)
        if self.output_encoder is not None and which is None:
            ys = self.output_encoder.encode(ys)
        self._x, self._y = xs, ys
        self.transform_x = PositionalEmbedding2D(grid_boundaries) if positional_encoding else None

    def _read(self, path: Path):
        """`.npy` holding a pickled dict {"x": [n, H, W] permeability, "y": [n, H, W] pressure}; the channel axis is inserted
        at `channel_dim`, the input is fp32 (darcyflow_dataset.py:253-264)."""
        data = np.load(path.as_posix(), allow_pickle=True).item()
        x = np.expand_dims(np.asarray(data["x"]), self.channel_dim).astype(np.float32)
        y = np.expand_dims(np.asarray(data["y"]), self.channel_dim).copy()
        return x, y

    def __len__(self):
        return self._x.shape[0]

    def __getitem__(self, index):
        x, y = self._x[index], self._y[index]
        if self.transform_x is not None:
            x = self.transform_x(x)
        return {self.input_keys[0]: x}, {self.label_keys[0]: y}, self.weight_dict
