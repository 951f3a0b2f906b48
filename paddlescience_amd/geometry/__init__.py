"""ppsci.geometry: numpy host-side geometries and samplers (bit-exact with the reference)."""
from .base import Geometry  # noqa: F401
from .boolean import CSGDifference, CSGIntersection, CSGUnion, Disk  # noqa: F401
from .pointcloud import PointCloud  # noqa: F401
from .polygon import Polygon, Triangle  # noqa: F401
from .shapes import Cuboid, Hypercube, Interval, Rectangle  # noqa: F401
from .timedomain import TimeDomain, TimeXGeometry  # noqa: F401

__all__ = ["Geometry", "Disk", "CSGUnion", "CSGDifference", "CSGIntersection", "PointCloud", "Triangle", "Polygon", "Interval", "Rectangle", "Cuboid", "Hypercube", "TimeDomain", "TimeXGeometry", "build_geometry"]


def build_geometry(cfg):
    """ppsci/geometry/__init__.py: cfg is a list of {name: {ClassName: kwargs}}; TimeXGeometry nests two of them."""
    if cfg is None:
        return None
    import copy

    out = {}
    for item in copy.deepcopy(cfg):
        name = next(iter(item.keys()))
        spec = item[name]
        cls = next(iter(spec.keys()))
        kwargs = spec[cls]
        if cls == "TimeXGeometry":
            td = TimeDomain(**kwargs.pop("TimeDomain"))
            gcls = next(iter(kwargs.keys()))
            out[name] = TimeXGeometry(td, globals()[gcls](**kwargs[gcls]))
        else:
            out[name] = globals()[cls](**kwargs)
    return out
