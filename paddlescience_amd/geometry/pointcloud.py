"""ppsci.geometry.PointCloud (/root/reference/ppsci/geometry/pointcloud.py:26-312): a geometry given by explicit
interior / boundary point sets with named coordinates (which may include non-spatial columns, e.g. a viscosity);
sampling is a draw without replacement from those sets (np.random.choice, same RNG stream as the reference)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from ..utils import misc
from .base import Geometry


class PointCloud(Geometry):
    def __init__(self, interior: Dict[str, np.ndarray], coord_keys: Tuple[str, ...],
                 boundary: Optional[Dict[str, np.ndarray]] = None,
                 boundary_normal: Optional[Dict[str, np.ndarray]] = None):
        self.interior = misc.convert_to_array(interior, coord_keys)
        self.len = self.interior.shape[0]
        self.boundary = None if boundary is None else misc.convert_to_array(boundary, coord_keys)
        self.normal = None
        if boundary_normal is not None:
            self.normal = misc.convert_to_array(boundary_normal, tuple(f"{k}_normal" for k in coord_keys))
            if list(self.normal.shape) != list(self.boundary.shape):
                raise ValueError(f"boundary's shape({self.boundary.shape}) must equal to normal's shape({self.normal.shape})")
        self.input_keys = coord_keys
        super().__init__(len(coord_keys), (np.amin(self.interior, axis=0), np.amax(self.interior, axis=0)), np.inf)

    @property
    def dim_keys(self):
        return self.input_keys

    @staticmethod
    def _member(x, pts):
        return np.isclose(x[:, None, :] - pts[None, :, :], 0, atol=1e-6).all(axis=2).any(axis=1)

    def is_inside(self, x):  # points of the boundary set count as inside when they are listed in `interior`
        return self._member(x, self.interior)

    def on_boundary(self, x):
        if self.boundary is None:
            raise ValueError("self.boundary must be initialized when call 'on_boundary' function")
        return self._member(x, self.boundary)

    def translate(self, translation: np.ndarray) -> "PointCloud":
        for i, offset in enumerate(translation):
            self.interior[:, i] += offset
            if self.boundary is not None:
                self.boundary += offset  # pointcloud.py:137-140 adds the offset to every boundary column
        return self

    def scale(self, scale: np.ndarray) -> "PointCloud":
        for i, factor in enumerate(scale):
            self.interior[:, i] *= factor
            if self.boundary is not None:
                self.boundary[:, i] *= factor
            if self.normal is not None:
                self.normal[:, i] *= factor
        return self

    def uniform_boundary_points(self, n: int):
        raise NotImplementedError("PointCloud do not have 'uniform_boundary_points' method")

    def random_boundary_points(self, n: int, random: str = "pseudo") -> np.ndarray:
        assert self.boundary is not None, "boundary points can't be empty when call 'random_boundary_points' method"
        assert n <= len(self.boundary), (f"number of sample points({n}) can't be more than that in "
                                         f"boundary({len(self.boundary)})")
        return self.boundary[np.random.choice(len(self.boundary), size=n, replace=False)]

    def random_points(self, n: int, random: str = "pseudo") -> np.ndarray:
        assert n <= len(self.interior), f"number of sample points({n}) can't be more than that in points({len(self.interior)})"
        return self.interior[np.random.choice(len(self.interior), size=n, replace=False)]

    def uniform_points(self, n: int, boundary: bool = True) -> np.ndarray:
        return self.interior[:n]

    def _no_csg(self, other):
        raise NotImplementedError("Boolean operations are not supported for PointCloud")

    union = __or__ = difference = __sub__ = intersection = __and__ = _no_csg

    def __str__(self) -> str:
        return ", ".join([self.__class__.__name__, f"num_points = {len(self.interior)}", f"ndim = {self.ndim}",
                          f"bbox = {self.bbox}", f"dim_keys = {self.dim_keys}"])
