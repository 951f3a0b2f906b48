"""ppsci.equation.NormalDotVec (/root/reference/ppsci/equation/pde/normal_dot_vec.py:22-59): n . v = 0 on a boundary; the normal
components `normal_x / _y / _z` are columns the boundary sampler provides."""
from typing import Optional, Tuple

from .base import PDE


class NormalDotVec(PDE):
    def __init__(self, vec_keys: Tuple[str, ...], detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        if not vec_keys:
            raise ValueError(f"len(vec_keys)({len(vec_keys)}) should be larger than 0.")
        self.vec_keys = vec_keys
        components = self.create_symbols(" ".join(vec_keys))
        if len(vec_keys) == 1:
            components = (components,)
        normals = self.create_symbols("normal_x normal_y normal_z")
        self.add_equation("normal_dot_vec", sum(n * c for n, c in zip(normals, components)))
        self._apply_detach()
