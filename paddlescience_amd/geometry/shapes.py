"""Axis-aligned shapes: Interval, Hypercube, Rectangle, Cuboid.

Numerics follow /root/reference/ppsci/geometry/geometry_1d.py:31-125, geometry_nd.py:33-135,
geometry_2d.py:103-238 (Rectangle) and geometry_3d.py:31-160 (Cuboid) operation by operation --
including the float32 grid step of Hypercube.uniform_points and the order in which numpy's global RNG
is consumed -- because sampled point sets have to be bit-identical to the reference's."""
from __future__ import annotations

import itertools
from typing import Tuple

import numpy as np

from ..utils import misc
from ..utils.misc import DEFAULT_DTYPE as F32
from .base import Geometry
from .sampler import sample


class Interval(Geometry):
    def __init__(self, l: float, r: float):
        super().__init__(1, (np.array([[l]]), np.array([[r]])), r - l)
        self.l, self.r = l, r

    def is_inside(self, x):
        return ((self.l <= x) & (x <= self.r)).flatten()

    def on_boundary(self, x):
        return (np.isclose(x, self.l) | np.isclose(x, self.r)).flatten()

    def boundary_normal(self, x):
        return -np.isclose(x, self.l).astype(F32) + np.isclose(x, self.r).astype(F32)

    def uniform_points(self, n: int, boundary: bool = True):
        if boundary:
            return np.linspace(self.l, self.r, n, dtype=F32).reshape([-1, 1])
        return np.linspace(self.l, self.r, n + 1, endpoint=False, dtype=F32)[1:].reshape([-1, 1])

    def random_points(self, n: int, random: str = "pseudo"):
        return (self.l + sample(n, 1, random) * self.diam).astype(F32)

    def uniform_boundary_points(self, n: int):
        if n == 1:
            return np.array([[self.l]], dtype=F32)
        return np.concatenate((np.full([n // 2, 1], self.l, dtype=F32), np.full([n - n // 2, 1], self.r, dtype=F32)), axis=0)

    def random_boundary_points(self, n: int, random: str = "pseudo"):
        if n == 2:
            return np.array([[self.l], [self.r]], dtype=F32)
        return np.random.choice([self.l, self.r], n).reshape([-1, 1]).astype(F32)

    def periodic_point(self, x, component: int = 0):
        arr = misc.convert_to_array(x, self.dim_keys)
        on_l, on_r = np.isclose(arr, self.l), None
        arr[on_l] = self.r
        on_r = np.isclose(arr, self.r)  # note: evaluated after the first assignment, like the reference
        arr[on_r] = self.l
        normal = self.boundary_normal(arr)
        return {**misc.convert_to_dict(arr, self.dim_keys),
                **misc.convert_to_dict(normal, [f"normal_{k}" for k in self.dim_keys])}

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        if points.shape[1] != self.ndim:
            raise ValueError(f"Shape of given points should be [*, {self.ndim}], but got {points.shape}")
        return -((self.r - self.l) / 2 - np.abs(points - (self.l + self.r) / 2))


class Hypercube(Geometry):
    def __init__(self, xmin: Tuple[float, ...], xmax: Tuple[float, ...]):
        if len(xmin) != len(xmax):
            raise ValueError("Dimensions of xmin and xmax do not match.")
        self.xmin = np.array(xmin, dtype=F32)
        self.xmax = np.array(xmax, dtype=F32)
        if np.any(self.xmin >= self.xmax):
            raise ValueError("xmin >= xmax")
        self.side_length = self.xmax - self.xmin
        super().__init__(len(xmin), (self.xmin, self.xmax), np.linalg.norm(self.side_length))
        self.volume = np.prod(self.side_length, dtype=F32)

    def is_inside(self, x):
        return np.logical_and(np.all(x >= self.xmin, axis=-1), np.all(x <= self.xmax, axis=-1))

    def on_boundary(self, x):
        edge = np.logical_or(np.any(np.isclose(x, self.xmin), axis=-1), np.any(np.isclose(x, self.xmax), axis=-1))
        return np.logical_and(self.is_inside(x), edge)

    def boundary_normal(self, x):
        nrm = -np.isclose(x, self.xmin).astype(F32) + np.isclose(x, self.xmax)
        corner = np.count_nonzero(nrm, axis=-1) > 1  # vertices/edges: average of the face normals
        if np.any(corner):
            nrm[corner] /= np.linalg.norm(nrm[corner], axis=-1, keepdims=True)
        return nrm

    def uniform_points(self, n, boundary=True):
        dx = (self.volume / n) ** (1 / self.ndim)  # float32 arithmetic, as in the reference
        axes = []
        for i in range(self.ndim):
            ni = int(np.ceil(self.side_length[i] / dx))
            if boundary:
                axes.append(np.linspace(self.xmin[i], self.xmax[i], num=ni, dtype=F32))
            else:
                axes.append(np.linspace(self.xmin[i], self.xmax[i], num=ni + 1, endpoint=False, dtype=F32)[1:])
        x = np.array(list(itertools.product(*axes)), dtype=F32)  # first axis varies slowest
        return x[0:n] if len(x) > n else x

    def random_points(self, n, random="pseudo"):
        return (self.xmax - self.xmin) * sample(n, self.ndim, random) + self.xmin

    def random_boundary_points(self, n, random="pseudo"):
        x = sample(n, self.ndim, random)
        pick = np.random.randint(self.ndim, size=n)  # the face: snap one coordinate to 0 or 1
        x[np.arange(n), pick] = np.round(x[np.arange(n), pick])
        return (self.xmax - self.xmin) * x + self.xmin

    def periodic_point(self, x, component):
        y = misc.convert_to_array(x, self.dim_keys)
        on_min = np.isclose(y[:, component], self.xmin[component])
        on_max = np.isclose(y[:, component], self.xmax[component])
        y[:, component][on_min] = self.xmax[component]
        y[:, component][on_max] = self.xmin[component]
        normal = self.boundary_normal(y)
        return {**misc.convert_to_dict(y, self.dim_keys),
                **misc.convert_to_dict(normal, [f"normal_{k}" for k in self.dim_keys])}


class Rectangle(Hypercube):
    def __init__(self, xmin, xmax):
        super().__init__(xmin, xmax)
        self.perimeter = 2 * np.sum(self.xmax - self.xmin)
        self.area = np.prod(self.xmax - self.xmin)

    def uniform_boundary_points(self, n):
        nx, ny = np.ceil(n / self.perimeter * (self.xmax - self.xmin)).astype(int)
        x0, y0, x1, y1 = self.xmin[0], self.xmin[1], self.xmax[0], self.xmax[1]

        def col(v, k):
            return np.full([k, 1], v, dtype=F32)

        bottom = np.hstack((np.linspace(x0, x1, nx, endpoint=False, dtype=F32).reshape([nx, 1]), col(y0, nx)))
        right = np.hstack((col(x1, ny), np.linspace(y0, y1, ny, endpoint=False, dtype=F32).reshape([ny, 1])))
        top = np.hstack((np.linspace(x0, x1, nx + 1, dtype=F32)[1:].reshape([nx, 1]), col(y1, nx)))
        left = np.hstack((col(x0, ny), np.linspace(y0, y1, ny + 1, dtype=F32)[1:].reshape([ny, 1])))
        x = np.vstack((bottom, right, top, left))
        return x[0:n] if len(x) > n else x

    def random_boundary_points(self, n, random="pseudo"):
        l1 = self.xmax[0] - self.xmin[0]
        l2 = l1 + self.xmax[1] - self.xmin[1]
        l3 = l2 + l1
        u = np.ravel(sample(n + 10, 1, random))
        u = u[~np.isclose(u, l1 / self.perimeter)]  # drop parameters that land on a corner
        u = u[~np.isclose(u, l3 / self.perimeter)]
        u = u[0:n]
        u *= self.perimeter
        pts = []
        for s in u:  # walk the perimeter counter-clockwise from (xmin, ymin)
            if s < l1:
                pts.append([self.xmin[0] + s, self.xmin[1]])
            elif s < l2:
                pts.append([self.xmax[0], self.xmin[1] + (s - l1)])
            elif s < l3:
                pts.append([self.xmax[0] - (s - l2), self.xmax[1]])
            else:
                pts.append([self.xmin[0], self.xmax[1] - (s - l3)])
        return np.vstack(pts)

    @staticmethod
    def is_valid(vertices):
        return (len(vertices) == 4
                and all(np.isclose(np.prod(vertices[(i + 1) % 4] - vertices[i]), 0) for i in range(4)))

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        if points.shape[1] != self.ndim:
            raise ValueError(f"Shape of given points should be [*, {self.ndim}], but got {points.shape}")
        center = (self.xmin + self.xmax) / 2
        d = np.abs(points - center) - np.array([self.xmax - self.xmin]) / 2
        return (np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(np.max(d, axis=1), 0)).reshape(-1, 1)


class Cuboid(Hypercube):
    def __init__(self, xmin, xmax):
        super().__init__(xmin, xmax)
        dx = self.xmax - self.xmin
        self.area = 2 * np.sum(dx * np.roll(dx, 2))

    def random_boundary_points(self, n, random="pseudo"):
        pts = []
        density = n / self.area
        rect = Rectangle(self.xmin[:-1], self.xmax[:-1])
        for z in [self.xmin[-1], self.xmax[-1]]:
            u = rect.random_points(int(np.ceil(density * rect.area)), random=random)
            pts.append(np.hstack((u, np.full((len(u), 1), z, dtype=F32))))
        rect = Rectangle(self.xmin[::2], self.xmax[::2])
        for y in [self.xmin[1], self.xmax[1]]:
            u = rect.random_points(int(np.ceil(density * rect.area)), random=random)
            pts.append(np.hstack((u[:, 0:1], np.full((len(u), 1), y, dtype=F32), u[:, 1:])))
        rect = Rectangle(self.xmin[1:], self.xmax[1:])
        for x in [self.xmin[0], self.xmax[0]]:
            u = rect.random_points(int(np.ceil(density * rect.area)), random=random)
            pts.append(np.hstack((np.full((len(u), 1), x, dtype=F32), u)))
        pts = np.vstack(pts)
        if len(pts) > n:
            return pts[np.random.choice(len(pts), size=n, replace=False)]
        return pts

    def uniform_boundary_points(self, n):
        h = (self.area / n) ** 0.5
        nx, ny, nz = np.ceil((self.xmax - self.xmin) / h).astype(int) + 1
        x = np.linspace(self.xmin[0], self.xmax[0], num=nx, dtype=F32)
        y = np.linspace(self.xmin[1], self.xmax[1], num=ny, dtype=F32)
        z = np.linspace(self.xmin[2], self.xmax[2], num=nz, dtype=F32)
        pts = []
        for v in [self.xmin[-1], self.xmax[-1]]:
            u = list(itertools.product(x, y))
            pts.append(np.hstack((u, np.full((len(u), 1), v, dtype=F32))))
        if nz > 2:
            for v in [self.xmin[1], self.xmax[1]]:
                u = np.array(list(itertools.product(x, z[1:-1])), dtype=F32)
                pts.append(np.hstack((u[:, 0:1], np.full((len(u), 1), v, dtype=F32), u[:, 1:])))
        if ny > 2 and nz > 2:
            for v in [self.xmin[0], self.xmax[0]]:
                u = list(itertools.product(y[1:-1], z[1:-1]))
                pts.append(np.hstack((np.full((len(u), 1), v, dtype=F32), u)))
        pts = np.vstack(pts)
        if len(pts) > n:
            return pts[np.random.choice(len(pts), size=n, replace=False)]
        return pts

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        if points.shape[1] != self.ndim:
            raise ValueError(f"Shape of given points should be [*, {self.ndim}], but got {points.shape}")
        sdf = ((self.xmax - self.xmin) / 2 - abs(points - (self.xmin + self.xmax) / 2)).min(axis=1)
        return -sdf[..., np.newaxis]
