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
    """darcyflow_dataset.py:168-296 (same constructor arguments and item layout)."""

    batch_index: bool = True

    def __init__(self, input_keys: Tuple[str, ...], label_keys: Tuple[str, ...], data_dir: str,
                 weight_dict: Optional[Dict[str, float]] = None, test_resolutions: Sequence[int] = (32,),
                 train_resolution: int = 32, grid_boundaries=((0, 1), (0, 1)), positional_encoding: bool = True,
                 encode_input: bool = False, encode_output: bool = True, encoding: str = "channel-wise",
                 channel_dim: int = 1, data_split: str = "train"):
        for res in test_resolutions:
            if res not in [16, 32]:
                raise ValueError(f"Only 32 and 64 are supported for test resolution, but got {test_resolutions}")
        self.input_keys, self.label_keys, self.data_dir = tuple(input_keys), tuple(label_keys), data_dir
        self.weight_dict = {} if weight_dict is None else weight_dict
        if weight_dict is not None:
            self.weight_dict = {key: 1.0 for key in self.label_keys}
            self.weight_dict.update(weight_dict)
        self.weight = self.weight_dict  # the array datasets' attribute name (Solver)
        self.test_resolutions, self.train_resolution = list(test_resolutions), train_resolution
        self.grid_boundaries, self.positional_encoding = grid_boundaries, positional_encoding
        self.encode_input, self.encode_output, self.encoding = encode_input, encode_output, encoding
        self.channel_dim, self.data_split = channel_dim, data_split

        self.x_train, self.y_train = self.read_data(Path(data_dir).joinpath(f"darcy_train_{train_resolution}.npy").as_posix())
        self.x_test_1, self.y_test_1 = self.read_data(
            Path(data_dir).joinpath(f"darcy_test_{self.test_resolutions[0]}.npy").as_posix())
        self.x_test_2, self.y_test_2 = self.read_data(
            Path(data_dir).joinpath(f"darcy_test_{self.test_resolutions[1]}.npy").as_posix())
        if self.encode_input:
            self.input_encoder = self.encode_data(self.x_train)
            self.x_train = self.input_encoder.encode(self.x_train)
            self.x_test_1 = self.input_encoder.encode(self.x_test_1)
            self.x_test_2 = self.input_encoder.encode(self.x_test_2)
        else:
            self.input_encoder = None
        if self.encode_output:
            self.output_encoder = self.encode_data(self.y_train)
            self.y_train = self.output_encoder.encode(self.y_train)
        else:
            self.output_encoder = None
        self.transform_x = PositionalEmbedding2D(grid_boundaries) if positional_encoding else None

    def read_data(self, path: str):
        data = np.load(path, allow_pickle=True).item()
        x = np.expand_dims(np.asarray(data["x"]), self.channel_dim).astype(np.float32)
        y = np.expand_dims(np.asarray(data["y"]), self.channel_dim).copy()
        return x, y

    def encode_data(self, data: np.ndarray) -> UnitGaussianNormalizer:
        if self.encoding == "channel-wise":
            reduce_dims = list(range(data.ndim))
        elif self.encoding == "pixel-wise":
            reduce_dims = [0]
        else:
            raise ValueError(f"encoding={self.encoding!r}")
        return UnitGaussianNormalizer(data, reduce_dim=reduce_dims)

    def _split(self):
        if self.data_split == "train":
            return self.x_train, self.y_train
        if self.data_split == "test_16x16":
            return self.x_test_1, self.y_test_1
        return self.x_test_2, self.y_test_2

    def __len__(self):
        return self._split()[0].shape[0]

    def __getitem__(self, index):
        xs, ys = self._split()
        x, y = xs[index], ys[index]
        if self.transform_x is not None:
            x = self.transform_x(x)
        return {self.input_keys[0]: x}, {self.label_keys[0]: y}, self.weight_dict
