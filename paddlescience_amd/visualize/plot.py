"""Scatter / image / trajectory plots (/root/reference/ppsci/visualize/plot.py:73-415), drawn with matplotlib when it is
importable.  Without it the arrays that would have been drawn go to `<filename>.npz` and one warning says so: an
example script still reaches its last line (the figures are not part of the hot path)."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from ..utils import logger

_COLORS = ("tab:blue", "tab:orange", "tab:green", "tab:red", "tab:purple", "tab:brown", "tab:pink", "tab:gray", "tab:olive",
           "tab:cyan")
_LINE_CMAPS = ("Greys", "Purples", "Blues", "Greens", "Oranges", "Reds", "YlOrBr", "YlOrRd", "OrRd", "PuRd", "RdPu", "BuPu",
               "GnBu", "PuBu", "YlGnBu", "PuBuGn", "BuGn", "YlGn")


def _np(a) -> np.ndarray:
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def _pyplot():
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt

        return plt
    except Exception:  # noqa: BLE001  (not installed, or no usable backend)
        return None


def _mkdir_for(filename: str) -> None:
    folder = os.path.dirname(filename)
    if folder:
        os.makedirs(folder, exist_ok=True)


def _fallback(filename: str, arrays: Dict[str, np.ndarray]) -> None:
    np.savez(filename + ".npz", **arrays)
    logger.warning(f"matplotlib is not available: the data of the figure is saved to {filename}.npz instead")


def _span(filename: str, kind: str, num_timestamps: int) -> None:
    if num_timestamps == 1:
        logger.message(f"{kind} result is saved to: {filename}.png")
    else:
        logger.message(f"{kind} result is saved to: {filename}_0.png ~ {filename}_{num_timestamps - 1}.png")


def save_plot_from_1d_dict(filename: str, data_dict: Dict[str, np.ndarray], coord_keys: Tuple[str, ...],
                           value_keys: Tuple[str, ...], num_timestamps: int = 1) -> None:
    """plot.py:124-168: one scatter panel per value key and timestamp over the (single) space coordinate."""
    space = [k for k in coord_keys if k != "t"]
    if len(space) not in (1, 2, 3):
        raise ValueError(f"ndim of space coord ({len(space)}) should be 1, 2 or 3")
    coord = np.concatenate([_np(data_dict[k]).reshape(len(_np(data_dict[k])), -1) for k in space], axis=1)
    value = np.concatenate([_np(data_dict[k]).reshape(len(coord), -1) for k in value_keys], axis=1) if value_keys else None
    _mkdir_for(filename)
    plt = _pyplot()
    if plt is None:
        return _fallback(filename, {"coord": coord, **({} if value is None else {k: value[:, i] for i, k in enumerate(value_keys)})})
    fig, axes = plt.subplots(len(value_keys), num_timestamps, squeeze=False)
    fig.subplots_adjust(hspace=0.8)
    per_t = len(coord) // num_timestamps
    for t in range(num_timestamps):
        sl = slice(t * per_t, (t + 1) * per_t)
        for i, key in enumerate(value_keys):
            ax = axes[i][t]
            ax.scatter(coord[sl, 0], value[sl, i], color=_COLORS[i % len(_COLORS)], label=key, s=2)
            ax.set_title(f"{key}(t={t})" if num_timestamps > 1 else f"{key}")
            ax.grid(color="#c2ccd0", linestyle="--", linewidth=0.5)
            ax.legend()
        fig.savefig(filename if num_timestamps == 1 else f"{filename}_{t}", dpi=300)
    plt.close(fig)
    _span(filename, "1D", num_timestamps)


def save_plot_from_2d_dict(filename: str, data_dict: Dict[str, np.ndarray], visu_keys: Tuple[str, ...], num_timestamps: int = 1,
                           stride: int = 1, xticks: Optional[Sequence[float]] = None,
                           yticks: Optional[Sequence[float]] = None) -> None:
    """plot.py:243-269: one row of images per key (`data[key][t * stride]`), a colour scale per row; rows whose key
    holds "target" set the scale of the prediction row that follows them."""
    fields = [_np(data_dict[k]) for k in visu_keys]
    _mkdir_for(filename)
    plt = _pyplot()
    if plt is None:
        return _fallback(filename, dict(zip(visu_keys, fields)))
    plt.close("all")
    fig, axes = plt.subplots(len(visu_keys), num_timestamps, squeeze=False, sharey=True, figsize=(max(num_timestamps, 2), max(len(visu_keys), 2)))
    fig.subplots_adjust(hspace=0.3)
    has_target = any("target" in k for k in visu_keys)
    extent = None
    if xticks is not None and yticks is not None:
        extent = [float(np.min(xticks)), float(np.max(xticks)), float(np.min(yticks)), float(np.max(yticks))]
    lo = hi = 0.0
    for i, field in enumerate(fields):
        if not has_target or "target" in visu_keys[i]:
            lo, hi = float(np.amin(field)), float(np.amax(field))
        for j in range(num_timestamps):
            t = j * stride
            im = axes[i, j].imshow(field[t], extent=extent, cmap="inferno", origin="lower", vmin=lo, vmax=hi)
            if xticks is not None:
                axes[i, j].set_xticks(list(xticks))
            if yticks is not None:
                axes[i, j].set_yticks(list(yticks))
            axes[i, j].tick_params(labelsize=5)
            axes[i, j].set_title(f"t={t}", fontsize=8)
        axes[i, 0].set_ylabel(visu_keys[i], fontsize=8)
        fig.colorbar(im, ax=list(axes[i]), fraction=0.02, pad=0.01).ax.tick_params(labelsize=5)
    fig.savefig(filename, dpi=300)
    plt.close(fig)
    logger.message(f"2D result is saved to: {filename}.png")


def save_plot_from_3d_dict(filename: str, data_dict: Dict[str, np.ndarray], visu_keys: Tuple[str, ...],
                           num_timestamps: int = 1) -> None:
    """plot.py:380-415: every key is a trajectory `[n, 3]`, drawn as a line in space coloured along its length."""
    tracks = [_np(data_dict[k]) for k in visu_keys]
    _mkdir_for(filename)
    plt = _pyplot()
    if plt is None:
        return _fallback(filename, dict(zip(visu_keys, tracks)))
    from matplotlib.lines import Line2D
    from mpl_toolkits.mplot3d.art3d import Line3DCollection

    fig = plt.figure(figsize=(10, 10))
    per_t = len(tracks[0]) // num_timestamps
    for t in range(num_timestamps):
        ax = fig.add_subplot(1, num_timestamps, t + 1, projection="3d")
        handles = []
        for i, track in enumerate(tracks):
            p = track[t * per_t:(t + 1) * per_t].reshape(-1, 3)
            cmap = plt.get_cmap(_LINE_CMAPS[i % len(_LINE_CMAPS)])
            if len(p) > 1:
                seg = np.stack([p[:-1], p[1:]], axis=1)
                lc = Line3DCollection(seg, cmap=cmap, linewidths=2)
                lc.set_array(np.linspace(0.2, 1.0, len(seg)))
                ax.add_collection3d(lc)
            ax.auto_scale_xyz(p[:, 0], p[:, 1], p[:, 2])
            handles.append(Line2D([0], [0], color=cmap(0.7), lw=2))
        ax.legend(handles, list(visu_keys), loc="upper right", framealpha=0.95)
        fig.savefig(filename if num_timestamps == 1 else f"{filename}_{t}", dpi=300)
    plt.close(fig)
    _span(filename, "3D", num_timestamps)
