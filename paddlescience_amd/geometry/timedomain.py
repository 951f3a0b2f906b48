"""TimeDomain and TimeXGeometry (/root/reference/ppsci/geometry/timedomain.py:39-783) for non-mesh
geometries: spatial points are drawn once and repeated over the time stamps (t0 excluded), the whole
set is truncated to n -- same order of RNG consumption as the reference."""
from __future__ import annotations

import itertools
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from ..utils import misc
from ..utils.misc import DEFAULT_DTYPE as F32
from .base import Geometry
from .shapes import Interval


class TimeDomain(Interval):
    def __init__(self, t0: float, t1: float, time_step: Optional[float] = None,
                 timestamps: Optional[Tuple[float, ...]] = None):
        super().__init__(t0, t1)
        self.t0, self.t1, self.time_step = t0, t1, time_step
        self.timestamps = None if timestamps is None else np.array(timestamps, dtype=F32).reshape([-1])
        if time_step is not None:
            if time_step <= 0:
                raise ValueError(f"time_step({time_step}) must be larger than 0.")
            self.num_timestamps = int(np.ceil((t1 - t0) / time_step)) + 1
        elif timestamps is not None:
            self.num_timestamps = len(timestamps)

    def on_initial(self, t: np.ndarray) -> np.ndarray:
        return np.isclose(t, self.t0).flatten()


class TimeXGeometry(Geometry):
    def __init__(self, timedomain: TimeDomain, geometry: Geometry):
        self.timedomain = timedomain
        self.geometry = geometry
        self.ndim = geometry.ndim + timedomain.ndim

    @property
    def dim_keys(self):
        return ("t",) + self.geometry.dim_keys

    def on_boundary(self, x):
        return self.geometry.on_boundary(x[:, 1:])

    def on_initial(self, x):
        return self.timedomain.on_initial(x[:, :1])

    def boundary_normal(self, x):
        return np.hstack((x[:, :1], self.geometry.boundary_normal(x[:, 1:])))

    # ---- helpers
    def _stamps(self, endpoint: bool = False):
        """Time levels after t0 and their count (timedomain.py:222-233, 447-456)."""
        td = self.timedomain
        if td.time_step is not None:
            nt = int(np.ceil(td.diam / td.time_step))
            return np.linspace(td.t1, td.t0, num=nt, endpoint=endpoint, dtype=F32)[:, None][::-1], nt
        return td.timestamps[1:], td.num_timestamps - 1

    def _space_fill(self, nx: int, draw: Callable[[], np.ndarray], criteria, what: str) -> np.ndarray:
        x = np.empty(shape=(nx, self.geometry.ndim), dtype=F32)
        size = tries = hits = 0
        while size < nx:
            pts = draw()
            if criteria is not None:  # the time argument is fixed to None (timedomain.py:247-251)
                pts = pts[criteria(None, *np.split(pts, self.geometry.ndim, axis=1)).flatten()]
            if len(pts) > nx - size:
                pts = pts[: nx - size]
            x[size: size + len(pts)] = pts
            size += len(pts)
            tries += 1
            hits += len(pts) > 0
            if tries >= 1000 and hits == 0:
                raise ValueError(f"Sample {what} failed, please check correctness of geometry and given criteria.")
        return x

    @staticmethod
    def _repeat(t, x, n):
        nx = len(x)
        tx = np.vstack([np.hstack((np.full([nx, 1], ti, dtype=F32), x)) for ti in t])
        return tx[:n] if len(tx) > n else tx

    # ---- interior
    def uniform_points(self, n: int, boundary: bool = True) -> np.ndarray:
        td = self.timedomain
        stamped = td.time_step is not None or td.timestamps is not None
        if td.time_step is not None:
            nt = int(np.ceil(td.diam / td.time_step))
            nx = int(np.ceil(n / nt))
        elif td.timestamps is not None:
            nt = td.num_timestamps - 1
            nx = int(np.ceil(n / nt))
        else:
            nx = int(np.ceil((n * np.prod(self.geometry.bbox[1] - self.geometry.bbox[0]) / td.diam) ** 0.5))
            nt = int(np.ceil(n / nx))
        x = self.geometry.uniform_points(nx, boundary=boundary)
        if boundary and not stamped:
            t = td.uniform_points(nt, boundary=True)
        elif td.time_step is not None:
            t = np.linspace(td.t1, td.t0, num=nt, endpoint=boundary, dtype=F32)[:, None][::-1]
        else:
            t = td.timestamps[1:]
        return self._repeat(t, x, n)

    def random_points(self, n: int, random: str = "pseudo", criteria: Optional[Callable] = None) -> np.ndarray:
        td = self.timedomain
        if td.time_step is None and td.timestamps is None:
            raise ValueError("Either time_step or timestamps must be provided.")
        t, nt = self._stamps()
        nx = int(np.ceil(n / nt))
        x = self._space_fill(nx, lambda: self.geometry.random_points(nx, random), criteria, "points")
        return self._repeat(t, x, n)

    # ---- boundary
    def uniform_boundary_points(self, n: int, criteria: Optional[Callable] = None) -> np.ndarray:
        td = self.timedomain
        if self.geometry.ndim == 1:
            nx = 2
        else:
            ext = self.geometry.bbox[1] - self.geometry.bbox[0]
            s = 2 * sum(a * b for a, b in itertools.combinations(ext, 2))
            nx = int((n * s / td.diam) ** 0.5)
        nt = int(np.ceil(n / nx))
        x = self._space_fill(nx, lambda: self.geometry.uniform_boundary_points(nx), criteria, "boundary points")
        t = np.linspace(td.t1, td.t0, num=nt, endpoint=False, dtype=F32)[:, None][::-1]
        return self._repeat(t, x, n)

    def random_boundary_points(self, n: int, random: str = "pseudo", criteria: Optional[Callable] = None) -> np.ndarray:
        td = self.timedomain
        if td.time_step is None and td.timestamps is None:
            raise ValueError("Either time_step or timestamps must be provided.")
        t, nt = self._stamps()
        nx = int(np.ceil(n / nt))
        x = self._space_fill(nx, lambda: self.geometry.random_boundary_points(nx, random), criteria, "boundary points")
        return self._repeat(t, x, n)

    # ---- initial
    def uniform_initial_points(self, n: int) -> np.ndarray:
        x = self.geometry.uniform_points(n, True)
        if len(x) > n:
            x = x[:n]
        return np.hstack((np.full([n, 1], self.timedomain.t0, dtype=F32), x))

    def random_initial_points(self, n: int, random: str = "pseudo") -> np.ndarray:
        x = self.geometry.random_points(n, random=random)
        return np.hstack((np.full([n, 1], self.timedomain.t0, dtype=F32), x))

    def sample_initial_interior(self, n: int, random: str = "pseudo", criteria: Optional[Callable] = None,
                                evenly: bool = False, compute_sdf_derivatives: bool = False) -> Dict[str, np.ndarray]:
        x = self._fill(n, (lambda: self.uniform_initial_points(n)) if evenly else (lambda: self.random_initial_points(n, random)),
                       criteria, 1000, "initial interior")
        out = misc.convert_to_dict(x, self.dim_keys)
        if hasattr(self.geometry, "sdf_func"):  # sdf of the spatial part only (timedomain.py:764-776)
            out.update(misc.convert_to_dict(-self.geometry.sdf_func(x[..., 1:]), ("sdf",)))
            if compute_sdf_derivatives:
                out.update(misc.convert_to_dict(-self.geometry.sdf_derivatives(x[..., 1:]),
                                                tuple(f"sdf__{k}" for k in self.geometry.dim_keys)))
        return out

    def periodic_point(self, x, component: int):
        xp = self.geometry.periodic_point({k: v for k, v in x.items() if k != "t"}, component)
        return {"t": x["t"], **xp}

    def __str__(self) -> str:
        return ", ".join([self.__class__.__name__, f"ndim = {self.ndim}", f"dim_keys = {self.dim_keys}"])
