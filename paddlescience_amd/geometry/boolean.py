"""Constructive solid geometry and the disk: ppsci.geometry.{Disk, CSGUnion, CSGDifference, CSGIntersection}
(/root/reference/ppsci/geometry/geometry_2d.py:32-106, ppsci/geometry/csg.py:25-337) and the `|`, `-`, `&`
operators of Geometry (geometry.py:520-660).  Host numpy; the draws and their order follow the reference so that
the same numpy RNG state gives bit-identical samples (tests/golden/geometry.npz).

The three boolean shapes differ only in (i) which part of each operand's boundary survives, (ii) the sign of the
second operand's normal, (iii) where interior candidates are drawn and how they are filtered, (iv) bbox / diameter
and (v) the SDF combination -- one class parametrised by a small table."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from ..utils.misc import DEFAULT_DTYPE
from . import sampler
from .base import Geometry


class Disk(Geometry):
    def __init__(self, center: Tuple[float, float], radius: float):
        self.center = np.array(center, dtype=DEFAULT_DTYPE)
        self.radius = radius
        super().__init__(2, (self.center - radius, self.center + radius), 2 * radius)

    def _dist(self, x, keepdims=False):
        return np.linalg.norm(x - self.center, axis=1, keepdims=keepdims)

    def is_inside(self, x):
        return self._dist(x) <= self.radius

    def on_boundary(self, x):
        return np.isclose(self._dist(x), self.radius)

    def boundary_normal(self, x):
        ox = x - self.center
        ox_len = np.linalg.norm(ox, axis=1, keepdims=True)
        return (ox / ox_len) * np.isclose(ox_len, self.radius).astype(DEFAULT_DTYPE)

    def random_points(self, n, random="pseudo"):
        rng = sampler.sample(n, 2, random)  # disk point picking: sqrt(r) along a uniform angle
        r, theta = rng[:, 0], 2 * np.pi * rng[:, 1]
        xy = np.stack((np.sqrt(r) * np.cos(theta), np.sqrt(r) * np.sin(theta)), axis=1)
        return self.radius * xy + self.center

    def uniform_boundary_points(self, n):
        theta = np.linspace(0, 2 * np.pi, num=n, endpoint=False, dtype=DEFAULT_DTYPE)
        return self.radius * np.stack((np.cos(theta), np.sin(theta)), axis=1) + self.center

    def random_boundary_points(self, n, random="pseudo"):
        theta = 2 * np.pi * sampler.sample(n, 1, random)
        return self.radius * np.concatenate((np.cos(theta), np.sin(theta)), axis=1) + self.center

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        if points.shape[1] != self.ndim:
            raise ValueError(f"Shape of given points should be [*, {self.ndim}], but got {points.shape}")
        return -(self.radius - self._dist(points))[..., np.newaxis]


# op -> (keep boundary-1 points where geom2.is_inside is ..., keep boundary-2 points where geom1.is_inside is ...,
#        sign of geom2's normal)
_BOUNDARY_RULE = {"union": (False, False, 1.0), "difference": (False, True, -1.0), "intersection": (True, True, 1.0)}


class BooleanGeometry(Geometry):
    op = "union"

    def __init__(self, geom1: Geometry, geom2: Geometry):
        if geom1.ndim != geom2.ndim:
            raise ValueError(f"{geom1}.ndim({geom1.ndim}) should be equal to {geom2}.ndim({geom1.ndim})")
        if self.op == "union":
            bbox = (np.minimum(geom1.bbox[0], geom2.bbox[0]), np.maximum(geom1.bbox[1], geom2.bbox[1]))
            diam = geom1.diam + geom2.diam
        elif self.op == "difference":
            bbox, diam = geom1.bbox, geom1.diam
        else:
            bbox = (np.maximum(geom1.bbox[0], geom2.bbox[0]), np.minimum(geom1.bbox[1], geom2.bbox[1]))
            diam = min(geom1.diam, geom2.diam)
        super().__init__(geom1.ndim, bbox, diam)
        self.geom1, self.geom2 = geom1, geom2

    # ---- membership
    def is_inside(self, x):
        a, b = self.geom1.is_inside(x), self.geom2.is_inside(x)
        if self.op == "union":
            return np.logical_or(a, b)
        return np.logical_and(a, ~b if self.op == "difference" else b)

    def _boundary_parts(self, x):
        """(points of geom1's boundary that bound the result, same for geom2)."""
        in2_wanted, in1_wanted, _ = _BOUNDARY_RULE[self.op]
        in2, in1 = self.geom2.is_inside(x), self.geom1.is_inside(x)
        part1 = np.logical_and(self.geom1.on_boundary(x), in2 if in2_wanted else ~in2)
        if self.op == "union":
            part2 = np.logical_and(self.geom2.on_boundary(x), ~in1)
        else:
            part2 = np.logical_and(in1, self.geom2.on_boundary(x))
        return part1, part2

    def on_boundary(self, x):
        return np.logical_or(*self._boundary_parts(x))

    def boundary_normal(self, x):
        part1, part2 = self._boundary_parts(x)
        n2 = self.geom2.boundary_normal(x)
        if _BOUNDARY_RULE[self.op][2] < 0:
            n2 = -n2
        return part1[:, np.newaxis] * self.geom1.boundary_normal(x) + part2[:, np.newaxis] * n2

    # ---- sampling (rejection; csg.py:66-104, :176-212, :287-323)
    def _collect(self, n, draw):
        x = np.empty(shape=(n, self.ndim), dtype=DEFAULT_DTYPE)
        size = 0
        while size < n:
            pts = draw()
            if len(pts) > n - size:
                pts = pts[: n - size]
            x[size: size + len(pts)] = pts
            size += len(pts)
        return x

    def random_points(self, n, random="pseudo"):
        def draw():
            if self.op == "union":  # uniform in the bounding box, kept where inside either operand
                pts = np.random.rand(n, self.ndim) * (self.bbox[1] - self.bbox[0]) + self.bbox[0]
                return pts[self.is_inside(pts)]
            pts = self.geom1.random_points(n, random=random)
            keep = self.geom2.is_inside(pts)
            return pts[~keep if self.op == "difference" else keep]

        return self._collect(n, draw)

    def random_boundary_points(self, n, random="pseudo"):
        in2_wanted, in1_wanted, _ = _BOUNDARY_RULE[self.op]

        def draw():
            b1 = self.geom1.random_boundary_points(n, random=random)
            k1 = self.geom2.is_inside(b1)
            b1 = b1[k1 if in2_wanted else ~k1]
            b2 = self.geom2.random_boundary_points(n, random=random)
            k2 = self.geom1.is_inside(b2)
            b2 = b2[k2 if in1_wanted else ~k2]
            return np.random.permutation(np.concatenate((b1, b2)))

        return self._collect(n, draw)

    def periodic_point(self, x, component):
        x = np.copy(x)
        part1, _ = self._boundary_parts(x)
        x[part1] = self.geom1.periodic_point(x, component)[part1]
        if self.op != "difference":  # the second mask is taken on the already updated points (csg.py:118-125)
            _, part2 = self._boundary_parts(x)
            x[part2] = self.geom2.periodic_point(x, component)[part2]
        return x

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        s1, s2 = self.geom1.sdf_func(points), self.geom2.sdf_func(points)
        if self.op == "union":
            return np.minimum(s1, s2)
        return np.maximum(s1, -s2 if self.op == "difference" else s2)


class CSGUnion(BooleanGeometry):
    op = "union"


class CSGDifference(BooleanGeometry):
    op = "difference"


class CSGIntersection(BooleanGeometry):
    op = "intersection"
