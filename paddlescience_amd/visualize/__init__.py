"""ppsci.visualize (/root/reference/ppsci/visualize/__init__.py:19-52): the visualizer classes the examples hand to
`Solver(visualizer=...)` and the `save_*_from_dict` writers.

Scope: SURVEY.md puts the reference's 1 451-line writer subsystem outside the hot path.  What lives here is the
API surface the example scripts touch -- constructors with the reference's argument order, `save(filename, data)`,
the `.vtu` point-cloud writer and the matplotlib scatter / image plots -- so that an unmodified example runs to its
last line.  The expressions of a visualizer are evaluated by `Solver.predict`, i.e. by the same compiled HIP
forward as a validator.  Not built: the weather and radar writers (they raise with the reason)."""
from __future__ import annotations

import copy

from .base import Visualizer
from .plot import save_plot_from_1d_dict, save_plot_from_2d_dict, save_plot_from_3d_dict
from .visualizer import (Visualizer2D, Visualizer2DPlot, Visualizer3D, VisualizerRadar, VisualizerScatter1D,
                         VisualizerScatter3D, VisualizerVtu, VisualizerWeather)
from .vtu import save_vtu_from_dict, save_vtu_to_mesh


def save_plot_weather_from_dict(*args, **kwargs):
    raise NotImplementedError("the weather writer (ppsci/visualize/plot.py:517-581) belongs to the data-driven model zoo: out of scope")


__all__ = ["Visualizer", "VisualizerScatter1D", "VisualizerScatter3D", "VisualizerVtu", "Visualizer2D", "Visualizer2DPlot",
           "Visualizer3D", "VisualizerWeather", "VisualizerRadar", "save_vtu_from_dict", "save_vtu_to_mesh",
           "save_plot_from_1d_dict", "save_plot_from_2d_dict", "save_plot_from_3d_dict", "save_plot_weather_from_dict",
           "build_visualizer"]

_CLASSES = {c.__name__: c for c in (VisualizerScatter1D, VisualizerScatter3D, VisualizerVtu, Visualizer2D, Visualizer2DPlot,
                                    Visualizer3D, VisualizerWeather, VisualizerRadar)}


def build_visualizer(cfg):
    """__init__.py:55-80: `[{ClassName: {kwargs}}, ...]` -> `{name: visualizer}`; names must be unique."""
    if cfg is None:
        return None
    out = {}
    for item in copy.deepcopy(cfg):
        cls_name = next(iter(item.keys()))
        kwargs = item[cls_name]
        if cls_name not in _CLASSES:
            raise ValueError(f"unknown visualizer class {cls_name!r}")
        name = kwargs.get("name", cls_name)
        if name in out:
            raise ValueError(f"Name of visualizer({name}) should be unique")
        out[name] = _CLASSES[cls_name](**kwargs)
    return out
