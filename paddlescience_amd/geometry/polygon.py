"""ppsci.geometry.{Triangle, Polygon} (/root/reference/ppsci/geometry/geometry_2d.py:236-452, :455-665).

Both are closed chains of straight edges, so both are written over one edge table (start vertex, edge vector,
length, unit tangent, outward normal); the operation order inside each formula follows the reference so that the
same numpy RNG state gives bit-identical samples (tests/golden/geometry.npz).  Host numpy, init / sampling time only.
"""
from __future__ import annotations

import numpy as np
from scipy import spatial

from ..utils.misc import DEFAULT_DTYPE
from . import sampler
from .base import Geometry
from .shapes import Rectangle


def polygon_signed_area(vertices) -> float:
    """Shoelace formula (geometry_2d.py:668-683): positive for counter-clockwise vertex order."""
    xs = np.array([v[0] for v in vertices] + [vertices[0][0]], dtype=DEFAULT_DTYPE)
    ys = np.array([v[1] for v in vertices] + [vertices[0][1]], dtype=DEFAULT_DTYPE)
    return 0.5 * (np.sum(xs[:-1] * ys[1:]) - np.sum(xs[1:] * ys[:-1]))


def clockwise_rotation_90(v):
    """(x, y) -> (y, -x) for a [2, N] (or [2]) array (geometry_2d.py:686-695)."""
    return np.array([v[1], -v[0]], dtype=DEFAULT_DTYPE)


def _cross(a, b):
    """z component of a x b for a [2] vector and [N, 2] points."""
    return a[0] * b[..., 1] - a[1] * b[..., 0]


def is_left(p0, p1, p2):
    """> 0 where the points p2 [N, 2] lie left of the line p0 -> p1 (geometry_2d.py:698-706)."""
    return _cross(p1 - p0, p2 - p0).reshape((-1, 1))


def _edge_points(n_total, perimeter, starts, vectors, lengths):
    """Evenly spaced points per edge, edge k getting ceil(n / perimeter * length_k) of them (end point excluded)."""
    density = n_total / perimeter
    rows = [np.linspace(0, 1, num=int(np.ceil(density * ln)), endpoint=False, dtype=DEFAULT_DTYPE)[:, None] * vec + st
            for st, vec, ln in zip(starts, vectors, lengths)]
    x = np.vstack(rows)
    return x[0:n_total] if len(x) > n_total else x


class Triangle(Geometry):
    """Vertices in either orientation; stored counter-clockwise."""

    def __init__(self, x1, x2, x3):
        self.area = polygon_signed_area([x1, x2, x3])
        if self.area < 0:  # clockwise input
            self.area = -self.area
            x2, x3 = x3, x2
        self.x1, self.x2, self.x3 = (np.array(v, dtype=DEFAULT_DTYPE) for v in (x1, x2, x3))
        self.v12, self.v23, self.v31 = self.x2 - self.x1, self.x3 - self.x2, self.x1 - self.x3
        self.l12, self.l23, self.l31 = (np.linalg.norm(v) for v in (self.v12, self.v23, self.v31))
        self.n12, self.n23, self.n31 = self.v12 / self.l12, self.v23 / self.l23, self.v31 / self.l31
        self.n12_normal, self.n23_normal, self.n31_normal = (clockwise_rotation_90(t) for t in (self.n12, self.n23, self.n31))
        self.perimeter = self.l12 + self.l23 + self.l31
        a, b, c = self.l12, self.l23, self.l31
        # the circumscribed circle's diameter, a b c / sqrt(p (a+b-c)(b+c-a)(c+a-b))
        diam = a * b * c / (self.perimeter * (a + b - c) * (b + c - a) * (c + a - b)) ** 0.5
        super().__init__(2, (np.minimum(x1, np.minimum(x2, x3)), np.maximum(x1, np.maximum(x2, x3))), diam)

    def _edges(self):
        return ((self.x1, self.x2, self.v12, self.l12, self.n12, self.n12_normal),
                (self.x2, self.x3, self.v23, self.l23, self.n23, self.n23_normal),
                (self.x3, self.x1, self.v31, self.l31, self.n31, self.n31_normal))

    def is_inside(self, x):
        side = np.stack([_cross(v, x - a) for a, _, v, *_ in self._edges()], axis=1)
        return ~(np.any(side > 0, axis=-1) & np.any(side < 0, axis=-1))

    def _vertex_dist(self, x, keepdims=False):
        return [np.linalg.norm(x - v, axis=-1, keepdims=keepdims) for v in (self.x1, self.x2, self.x3)]

    def on_boundary(self, x):
        l1, l2, l3 = self._vertex_dist(x)
        return np.any(np.isclose([l1 + l2 - self.l12, l2 + l3 - self.l23, l3 + l1 - self.l31], 0, atol=1e-6), axis=0)

    def boundary_normal(self, x):
        l1, l2, l3 = self._vertex_dist(x, keepdims=True)
        on = [np.isclose(l1 + l2, self.l12), np.isclose(l2 + l3, self.l23), np.isclose(l3 + l1, self.l31)]
        if np.any(np.count_nonzero(np.hstack(on), axis=-1) > 1):
            raise ValueError(f"{self.__class__.__name__}.boundary_normal do not accept points on the vertexes.")
        return self.n12_normal * on[0] + self.n23_normal * on[1] + self.n31_normal * on[2]

    def random_points(self, n, random="pseudo"):
        # barycentric picking with a square-rooted first draw; numpy's global RNG directly, two [n, 1] draws
        sqrt_r1 = np.sqrt(np.random.rand(n, 1))
        r2 = np.random.rand(n, 1)
        return (1 - sqrt_r1) * self.x1 + sqrt_r1 * (1 - r2) * self.x2 + r2 * sqrt_r1 * self.x3

    def uniform_boundary_points(self, n):
        e = self._edges()
        return _edge_points(n, self.perimeter, [k[0] for k in e], [k[2] for k in e], [k[3] for k in e])

    def random_boundary_points(self, n, random="pseudo"):
        u = np.ravel(sampler.sample(n + 2, 1, random))
        for corner in (self.l12 / self.perimeter, (self.l12 + self.l23) / self.perimeter):  # drop draws at a corner
            u = u[np.logical_not(np.isclose(u, corner))]
        u = u[:n]
        u *= self.perimeter
        pts = []
        for s in u:
            if s < self.l12:
                pts.append(s * self.n12 + self.x1)
            elif s < self.l12 + self.l23:
                pts.append((s - self.l12) * self.n23 + self.x2)
            else:
                pts.append((s - self.l12 - self.l23) * self.n31 + self.x3)
        return np.vstack(pts)

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        if points.shape[1] != self.ndim:
            raise ValueError(f"Shape of given points should be [*, {self.ndim}], but got {points.shape}")
        dist = []
        for a, _, v, ln, *_ in self._edges():
            ap = points - a
            foot = v * np.clip(np.dot(ap, v.reshape(2, -1)) / ln ** 2, 0, 1) - ap  # point -> nearest point of the edge
            dist.append(np.linalg.norm(foot, axis=1, keepdims=True))
        sign = self.is_inside(points).reshape(-1, 1) * 2 - 1
        return sign * np.minimum(np.minimum(dist[0], dist[1]), dist[2])


class Polygon(Geometry):
    """Simple polygon with >= 4 vertices that is not an axis-aligned rectangle; vertices stored counter-clockwise."""

    def __init__(self, vertices):
        self.vertices = np.array(vertices, dtype=DEFAULT_DTYPE)
        if len(vertices) == 3:
            raise ValueError("The polygon is a triangle. Use Triangle instead.")
        if Rectangle.is_valid(self.vertices):
            raise ValueError("The polygon is a rectangle. Use Rectangle instead.")
        self.area = polygon_signed_area(self.vertices)
        if self.area < 0:
            self.area = -self.area
            self.vertices = np.flipud(self.vertices)
        self.diagonals = spatial.distance.squareform(spatial.distance.pdist(self.vertices))
        lo, hi = np.amin(self.vertices, axis=0), np.amax(self.vertices, axis=0)
        super().__init__(2, (lo, hi), np.max(self.diagonals))
        self.nvertices = len(self.vertices)
        self._edge_ids = range(-1, self.nvertices - 1)  # edge i runs from vertex i to vertex i + 1, starting at the last
        self.perimeter = np.sum([self.diagonals[i, i + 1] for i in self._edge_ids])
        self.bbox = np.array([lo, hi], dtype=DEFAULT_DTYPE)
        self.segments = np.vstack((self.vertices[0] - self.vertices[-1], self.vertices[1:] - self.vertices[:-1]))
        normal = clockwise_rotation_90(self.segments.T).T
        self.normal = normal / np.linalg.norm(normal, axis=1).reshape(-1, 1)

    def is_inside(self, x):
        """Winding number != 0 (upward crossings with the point on the left minus downward ones on the right)."""
        V, py = self.vertices, x[:, 1:2]
        wn = np.zeros(len(x))
        for i in self._edge_ids:
            side = is_left(V[i], V[i + 1], x)
            wn[np.all(np.hstack([V[i, 1] <= py, V[i + 1, 1] > py, side > 0]), axis=-1)] += 1
            wn[np.all(np.hstack([V[i, 1] > py, V[i + 1, 1] <= py, side < 0]), axis=-1)] -= 1
        return wn != 0

    def on_boundary(self, x):
        hits = np.zeros(shape=len(x), dtype=int)
        for i in self._edge_ids:
            l1 = np.linalg.norm(self.vertices[i] - x, axis=-1)
            l2 = np.linalg.norm(self.vertices[i + 1] - x, axis=-1)
            hits[np.isclose(l1 + l2, self.diagonals[i, i + 1])] += 1
        return hits > 0

    def random_points(self, n, random="pseudo"):
        x = np.empty((0, 2), dtype=DEFAULT_DTYPE)
        extent = self.bbox[1] - self.bbox[0]
        while len(x) < n:  # rejection from the bounding box, always pseudo-random
            cand = sampler.sample(n, 2, "pseudo") * extent + self.bbox[0]
            x = np.vstack((x, cand[self.is_inside(cand)]))
        return x[:n]

    def uniform_boundary_points(self, n):
        ids = list(self._edge_ids)
        return _edge_points(n, self.perimeter, [self.vertices[i] for i in ids],
                            [self.vertices[i + 1] - self.vertices[i] for i in ids], [self.diagonals[i, i + 1] for i in ids])

    def random_boundary_points(self, n, random="pseudo"):
        u = np.ravel(sampler.sample(n + self.nvertices, 1, random))
        run = 0
        for i in range(0, self.nvertices - 1):  # drop draws that land on a corner
            run += self.diagonals[i, i + 1]
            u = u[np.logical_not(np.isclose(u, run / self.perimeter))]
        u = u[:n]
        u *= self.perimeter
        u.sort()
        pts, i, start = [], -1, 0
        end = start + self.diagonals[i, i + 1]
        tangent = (self.vertices[i + 1] - self.vertices[i]) / self.diagonals[i, i + 1]
        for s in u:
            if s > end:  # the sorted arc length walks the edges once
                i += 1
                start, end = end, end + self.diagonals[i, i + 1]
                tangent = (self.vertices[i + 1] - self.vertices[i]) / self.diagonals[i, i + 1]
            pts.append((s - start) * tangent + self.vertices[i])
        return np.vstack(pts)

    def sdf_func(self, points: np.ndarray) -> np.ndarray:
        if points.shape[1] != self.ndim:
            raise ValueError(f"Shape of given points should be [*, {self.ndim}], but got {points.shape}")
        V, nv = self.vertices, self.vertices.shape[0]
        out = np.empty((points.shape[0], 1), dtype=DEFAULT_DTYPE)
        for n, p in enumerate(points):
            d0 = p - V[0]
            dist2, sign = np.dot(d0, d0), 1.0
            for i in range(nv):
                j = i - 1 if i else nv - 1
                edge, rel = V[j] - V[i], p - V[i]
                off = rel - edge * np.clip(np.dot(rel, edge) / np.dot(edge, edge), 0.0, 1.0)
                dist2 = np.minimum(dist2, np.dot(off, off))
                # even-odd crossing test of the ray through p
                tests = np.array([p[1] >= V[i][1], p[1] < V[j][1], edge[0] * rel[1] > edge[1] * rel[0]])
                if tests.all() or np.all(~tests):
                    sign *= -1.0
            out[n] = sign * np.sqrt(dist2)
        return -out
