import copy

from .array_dataset import ContinuousNamedArrayDataset, IterableNamedArrayDataset, NamedArrayDataset  # noqa: F401
from .darcyflow_dataset import DarcyFlowDataset  # noqa: F401
from .spherical_swe_dataset import SphericalSWEDataset  # noqa: F401
from .file_dataset import (CSVDataset, IterableCSVDataset, IterableMatDataset, IterableNPZDataset, MatDataset,  # noqa: F401
                           NPZDataset)

__all__ = ["NamedArrayDataset", "IterableNamedArrayDataset", "ContinuousNamedArrayDataset", "DarcyFlowDataset", "SphericalSWEDataset", "CSVDataset",
           "IterableCSVDataset", "MatDataset", "IterableMatDataset", "NPZDataset", "IterableNPZDataset", "build_dataset"]


def build_dataset(cfg):
    """ppsci/data/dataset/__init__.py: cfg = {"name": ClassName, **kwargs}."""
    cfg = dict(cfg)
    cls = cfg.pop("name")
    if cls not in __all__[:-1]:
        raise NotImplementedError(f"dataset {cls!r} (file-backed data-driven datasets are out of scope of the PINN hot path)")
    cfg.pop("transforms", None) if cfg.get("transforms") is None else None
    return globals()[cls](**cfg)
