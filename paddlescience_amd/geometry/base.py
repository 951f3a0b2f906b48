"""Geometry base class: rejection-loop samplers that return named [n,1] arrays
(/root/reference/ppsci/geometry/geometry.py:34-486).  Host-side numpy; results are bit-exact with the
reference for the same numpy global-RNG state (tests/test_geometry.py pins the reference doctests)."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

from ..utils import misc
from ..utils.misc import DEFAULT_DTYPE


class Geometry:
    def __init__(self, ndim: int, bbox: Tuple[np.ndarray, np.ndarray], diam: float):
        self.ndim = ndim
        self.bbox = bbox
        self.diam = min(diam, np.linalg.norm(bbox[1] - bbox[0]))

    @property
    def dim_keys(self):
        return ("x", "y", "z")[: self.ndim]

    # ---- to be provided by shapes
    def is_inside(self, x):
        raise NotImplementedError

    def on_boundary(self, x):
        raise NotImplementedError

    def boundary_normal(self, x):
        raise NotImplementedError(f"{self}.boundary_normal is not implemented")

    def random_points(self, n, random="pseudo"):
        raise NotImplementedError

    def random_boundary_points(self, n, random="pseudo"):
        raise NotImplementedError

    def uniform_points(self, n: int, boundary: bool = True):
        return self.random_points(n)

    def uniform_boundary_points(self, n: int):
        return self.random_boundary_points(n)

    def periodic_point(self, x, component):
        raise NotImplementedError(f"{self}.periodic_point to be implemented")

    # ---- samplers
    def _fill(self, n: int, draw: Callable[[], np.ndarray], criteria, limit: int, what: str) -> np.ndarray:
        """The reference's rejection loop (geometry.py:184-213, 282-323): draw n candidates, filter, keep what fits."""
        out = np.empty(shape=(n, self.ndim), dtype=DEFAULT_DTYPE)
        size = tries = hits = 0
        while size < n:
            pts = draw()
            if criteria is not None:
                pts = pts[criteria(*np.split(pts, self.ndim, axis=1)).flatten()]
            if len(pts) > n - size:
                pts = pts[: n - size]
            out[size: size + len(pts)] = pts
            size += len(pts)
            tries += 1
            hits += len(pts) > 0
            if tries >= limit and hits == 0:
                raise ValueError(f"Sample {what} points failed, please check correctness of geometry and given criteria.")
        return out

    def _is_time_x(self) -> bool:
        return misc.typename(self) == "TimeXGeometry"

    def sample_interior(self, n: int, random: str = "pseudo", criteria: Optional[Callable] = None, evenly: bool = False,
                        compute_sdf_derivatives: bool = False) -> Dict[str, np.ndarray]:
        def draw():
            if evenly:
                return self.uniform_points(n)
            if self._is_time_x():
                return self.random_points(n, random, criteria)
            return self.random_points(n, random)

        x = self._fill(n, draw, criteria, 1000, "interior")
        out = misc.convert_to_dict(x, self.dim_keys)
        if hasattr(self, "sdf_func"):
            out.update(misc.convert_to_dict(-self.sdf_func(x), ("sdf",)))
            if compute_sdf_derivatives:
                out.update(misc.convert_to_dict(-self.sdf_derivatives(x), tuple(f"sdf__{k}" for k in self.dim_keys)))
        return out

    def sample_boundary(self, n: int, random: str = "pseudo", criteria: Optional[Callable] = None,
                        evenly: bool = False) -> Dict[str, np.ndarray]:
        def draw():
            if evenly:
                return self.uniform_boundary_points(n)
            if self._is_time_x():
                return self.random_boundary_points(n, random, criteria)
            return self.random_boundary_points(n, random)

        x = self._fill(n, draw, criteria, 10000, "boundary")
        normal = self.boundary_normal(x)
        normal_dict = misc.convert_to_dict(normal[:, 1:] if "t" in self.dim_keys else normal,
                                           [f"normal_{k}" for k in self.dim_keys if k != "t"])
        return {**misc.convert_to_dict(x, self.dim_keys), **normal_dict}

    def sdf_derivatives(self, x: np.ndarray, epsilon: float = 1e-4) -> np.ndarray:
        """Central differences of sdf_func (geometry.py:439-486)."""
        if not hasattr(self, "sdf_func"):
            raise NotImplementedError(f"{misc.typename(self)}.sdf_func should be implemented when using 'sdf_derivatives'.")
        out = np.empty_like(x)
        for i in range(self.ndim):
            h = np.zeros_like(x)
            h[:, i] += epsilon / 2
            out[:, i: i + 1] = (self.sdf_func(x + h) - self.sdf_func(x - h)) / epsilon
        return out

    # ---- CSG (geometry.py:520-660)
    def union(self, other: "Geometry") -> "Geometry":
        from .boolean import CSGUnion

        return CSGUnion(self, other)

    def difference(self, other: "Geometry") -> "Geometry":
        from .boolean import CSGDifference

        return CSGDifference(self, other)

    def intersection(self, other: "Geometry") -> "Geometry":
        from .boolean import CSGIntersection

        return CSGIntersection(self, other)

    __or__ = __add__ = union
    __sub__ = difference
    __and__ = intersection

    def __str__(self) -> str:
        return ", ".join([self.__class__.__name__, f"ndim = {self.ndim}", f"bbox = {self.bbox}", f"diam = {self.diam}",
                          f"dim_keys = {self.dim_keys}"])
