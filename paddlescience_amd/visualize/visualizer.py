"""The visualizer classes of /root/reference/ppsci/visualize/visualizer.py:29-409 (constructor argument order and
defaults kept, so positional calls of the examples bind the same way)."""
from __future__ import annotations

import os.path as osp
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from . import plot, vtu
from .base import Visualizer


class VisualizerScatter1D(Visualizer):
    """visualizer.py:29-73: scatter of every output over the coordinate keys."""

    def __init__(self, input_dict: Dict[str, np.ndarray], coord_keys: Tuple[str, ...], output_expr: Dict[str, Callable],
                 batch_size: int = 64, num_timestamps: int = 1, prefix: str = "plot"):
        super().__init__(input_dict, output_expr, batch_size, num_timestamps, prefix)
        self.coord_keys = coord_keys

    def save(self, filename, data_dict):
        plot.save_plot_from_1d_dict(filename, data_dict, self.coord_keys, self.output_keys, self.num_timestamps)


class VisualizerScatter3D(Visualizer):
    """visualizer.py:76-128: trajectories in space; a leading sample axis gives one figure per sample."""

    def __init__(self, input_dict: Dict[str, np.ndarray], output_expr: Dict[str, Callable], batch_size: int = 64,
                 num_timestamps: int = 1, prefix: str = "plot3d_scatter"):
        super().__init__(input_dict, output_expr, batch_size, num_timestamps, prefix)

    def save(self, filename, data_dict):
        data = {k: np.asarray(v) for k, v in data_dict.items() if k in self.output_keys}
        if data[self.output_keys[0]].ndim == 3:
            for i in range(len(data[self.output_keys[0]])):
                plot.save_plot_from_3d_dict(filename + str(i), {k: v[i] for k, v in data.items()}, self.output_keys,
                                            self.num_timestamps)
        else:
            plot.save_plot_from_3d_dict(filename, data, self.output_keys, self.num_timestamps)


class VisualizerVtu(Visualizer):
    """visualizer.py:131-172: the input columns are the coordinates of a `.vtu` point cloud."""

    def __init__(self, input_dict: Dict[str, np.ndarray], output_expr: Dict[str, Callable], batch_size: int = 64,
                 num_timestamps: int = 1, prefix: str = "vtu"):
        super().__init__(input_dict, output_expr, batch_size, num_timestamps, prefix)

    def save(self, filename, data_dict):
        vtu.save_vtu_from_dict(filename, data_dict, self.input_keys, self.output_keys, self.num_timestamps)


class Visualizer2D(Visualizer):
    """visualizer.py:175-213."""

    def __init__(self, input_dict: Dict[str, np.ndarray], output_expr: Dict[str, Callable], batch_size: int = 64,
                 num_timestamps: int = 1, prefix: str = "plot2d"):
        super().__init__(input_dict, output_expr, batch_size, num_timestamps, prefix)


class Visualizer2DPlot(Visualizer2D):
    """visualizer.py:216-297: image rows over time; a leading sample axis gives one figure per sample."""

    def __init__(self, input_dict: Dict[str, np.ndarray], output_expr: Dict[str, Callable], batch_size: int = 64,
                 num_timestamps: int = 1, stride: int = 1, xticks: Optional[Tuple[float, ...]] = None,
                 yticks: Optional[Tuple[float, ...]] = None, prefix: str = "plot2d"):
        super().__init__(input_dict, output_expr, batch_size, num_timestamps, prefix)
        self.stride = stride
        self.xticks = xticks
        self.yticks = yticks

    def save(self, filename, data_dict):
        data = {k: np.asarray(v) for k, v in data_dict.items() if k in self.output_keys}
        args = (self.output_keys, self.num_timestamps, self.stride, self.xticks, self.yticks)
        if data[self.output_keys[0]].ndim == 4:
            for i in range(len(data[self.output_keys[0]])):
                plot.save_plot_from_2d_dict(filename + str(i), {k: v[i] for k, v in data.items()}, *args)
        else:
            plot.save_plot_from_2d_dict(filename, data, *args)


class Visualizer3D(Visualizer):
    """visualizer.py:300-338: one `predict_<i>.vtu` per entry of `time_list`."""

    def __init__(self, input_dict: Dict[str, np.ndarray], output_expr: Dict[str, Callable], batch_size: int = 64,
                 label_dict: Optional[Dict[str, np.ndarray]] = None, time_list: Optional[Tuple[float, ...]] = None,
                 prefix: str = "vtu"):
        self.label = label_dict
        self.time_list = time_list
        super().__init__(input_dict, output_expr, batch_size, len(time_list), prefix)

    def save(self, filename: str, data_dict: Dict[str, np.ndarray]):
        per_t = len(next(iter(data_dict.values()))) // self.num_timestamps
        coord_keys = [k for k in self.input_dict if k != "t"]
        for i in range(len(self.time_list)):
            vtu.save_vtu_to_mesh(osp.join(filename, f"predict_{i + 1}.vtu"),
                                 {k: v[i * per_t:(i + 1) * per_t] for k, v in data_dict.items()}, coord_keys, self.output_keys)


class VisualizerWeather(Visualizer):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("VisualizerWeather (visualizer.py:341-409) belongs to the data-driven weather models: out of scope")


class VisualizerRadar(Visualizer):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("VisualizerRadar (ppsci/visualize/radar.py) belongs to the data-driven radar models: out of scope")
