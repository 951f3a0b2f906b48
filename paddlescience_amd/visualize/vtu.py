"""`.vtu` point-cloud writers (/root/reference/ppsci/visualize/vtu.py:28-192).

The reference goes through pyevtk (`hl.pointsToVTK`) and meshio; neither is a dependency here.  A point cloud is a
VTK UnstructuredGrid whose cells are one VTK_VERTEX per point, written as ASCII XML (every VTK reader takes it):
same file names (`<name>.vtu`, `<name>_t-<k>.vtu` for several timestamps), same array names."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from ..utils import logger


def _np(a) -> np.ndarray:
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def _fmt(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    if a.dtype.kind in "iu":
        return " ".join(map(str, a.ravel().tolist()))
    return " ".join(f"{v:.9g}" for v in a.ravel().astype(np.float64).tolist())


def _write_points(path: str, xyz: np.ndarray, point_data: Dict[str, np.ndarray]) -> None:
    """One UnstructuredGrid piece: `xyz` [n, 3], every value array [n] or [n, c]."""
    n = len(xyz)
    rows = ['<?xml version="1.0"?>',
            '<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">',
            "<UnstructuredGrid>", f'<Piece NumberOfPoints="{n}" NumberOfCells="{n}">',
            "<Points>", '<DataArray type="Float32" NumberOfComponents="3" format="ascii">',
            _fmt(xyz.astype(np.float32)), "</DataArray>", "</Points>",
            "<Cells>", '<DataArray type="Int32" Name="connectivity" format="ascii">', _fmt(np.arange(n, dtype=np.int32)),
            "</DataArray>", '<DataArray type="Int32" Name="offsets" format="ascii">',
            _fmt(np.arange(1, n + 1, dtype=np.int32)), "</DataArray>",
            '<DataArray type="UInt8" Name="types" format="ascii">', _fmt(np.ones(n, dtype=np.uint8)), "</DataArray>",
            "</Cells>", "<PointData>"]
    for key, val in point_data.items():
        val = np.asarray(val, dtype=np.float32).reshape(n, -1)
        rows += [f'<DataArray type="Float32" Name="{key}" NumberOfComponents="{val.shape[1]}" format="ascii">',
                 _fmt(val), "</DataArray>"]
    rows += ["</PointData>", "</Piece>", "</UnstructuredGrid>", "</VTKFile>"]
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


def _save_vtu_from_array(filename: str, coord: np.ndarray, value: Optional[np.ndarray], value_keys: Sequence[str],
                         num_timestamps: int = 1) -> None:
    """vtu.py:28-111: the argument checks and the file naming of the reference."""
    if not isinstance(coord, np.ndarray):
        raise ValueError(f"type of coord({type(coord)}) should be ndarray.")
    if value is not None and not isinstance(value, np.ndarray):
        raise ValueError(f"type of value({type(value)}) should be ndarray.")
    if value is not None and len(coord) != len(value):
        raise ValueError(f"coord length({len(coord)}) should be equal to value length({len(value)})")
    if len(coord) % num_timestamps != 0:
        raise ValueError(f"coord length({len(coord)}) should be an integer multiple of num_timestamps({num_timestamps})")
    if coord.shape[1] not in (2, 3):
        raise ValueError(f"ndim of coord({coord.shape[1]}) should be 2 or 3.")
    folder = os.path.dirname(filename)
    if folder:
        os.makedirs(folder, exist_ok=True)
    if filename.endswith(".vtu"):
        filename = filename[:-4]
    if value is None:
        value, value_keys = np.ones((len(coord), 1), dtype=coord.dtype), ["dummy_key"]
    per_t = len(coord) // num_timestamps
    width = len(str(num_timestamps - 1))
    for t in range(num_timestamps):
        sl = slice(t * per_t, (t + 1) * per_t)
        xyz = np.zeros((per_t, 3), dtype=np.float32)
        xyz[:, :coord.shape[1]] = coord[sl]
        data = {key: value[sl, j] for j, key in enumerate(value_keys)}
        _write_points(f"{filename}_t-{t:0{width}}.vtu" if num_timestamps > 1 else f"{filename}.vtu", xyz, data)
    if num_timestamps > 1:
        logger.message(f"Visualization results are saved to: {filename}_t-{0:0{width}}.vtu ~ "
                       f"{filename}_t-{num_timestamps - 1:0{width}}.vtu")
    else:
        logger.message(f"Visualization result is saved to: {filename}.vtu")


def save_vtu_from_dict(filename: str, data_dict: Dict[str, np.ndarray], coord_keys: Tuple[str, ...],
                       value_keys: Tuple[str, ...], num_timestamps: int = 1) -> None:
    """vtu.py:114-155: columns of `data_dict` -> point cloud; "t" and "sdf" are not spatial coordinates."""
    if len(coord_keys) not in (2, 3, 4):
        raise ValueError(f"ndim of coord ({len(coord_keys)}) should be 2, 3 or 4")
    coord = np.concatenate([_np(data_dict[k]).reshape(len(_np(data_dict[k])), -1) for k in coord_keys if k not in ("t", "sdf")], axis=1)
    value = None
    if value_keys:
        value = np.concatenate([_np(data_dict[k]).reshape(len(coord), -1) for k in value_keys], axis=1)
    _save_vtu_from_array(filename, coord, value, value_keys, num_timestamps)


def save_vtu_to_mesh(filename: str, data_dict: Dict[str, np.ndarray], coord_keys: Tuple[str, ...],
                     value_keys: Tuple[str, ...]) -> None:
    """vtu.py:158-192: one vertex-cell file with the named point arrays."""
    n = len(_np(next(iter(data_dict.values()))))
    xyz = np.zeros((n, 3), dtype=np.float32)
    for j, key in enumerate(coord_keys):
        xyz[:, j] = _np(data_dict[key]).reshape(n)
    folder = os.path.dirname(filename)
    if folder:
        os.makedirs(folder, exist_ok=True)
    _write_points(filename, xyz, {key: _np(data_dict[key]) for key in value_keys})
