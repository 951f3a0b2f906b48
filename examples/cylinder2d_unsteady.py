"""2-D unsteady flow around a cylinder, Re = 100 -- /root/reference/examples/cylinder/2d_unsteady/cylinder2d_unsteady_Re100.py
(+ conf/cylinder2d_unsteady_Re100.yaml): MLP (t, x, y) -> (u, v, p), 5 x 50 tanh; NavierStokes(nu 0.02, rho 1, dim 2, time);
per iteration ONE full batch of every constraint: 9 420 domain points x 30 time stamps for the PDE, 161 inlet + cylinder
points and 81 outlet points x 30 time stamps, 9 420 initial-condition points, weights 10 on the velocity labels.  This is the
case behind the reference's published TIPC throughput (`ips` 1 264 165.6 points/s, fp32, one unnamed NVIDIA GPU,
test_tipc/README.MD:17): `ips` in the [Train] log lines below is the same quantity (train.py:106, printer.py:66).

The reference reads the point sets and OpenFOAM labels from ./datasets/*.csv (download_dataset.py; no network here): when
`data_dir` does not hold them, point sets of the same sizes are generated (uniform points of the channel [-8, 25] x [-8, 8] minus
the cylinder of radius 0.5, inlet / cylinder / outlet boundary points) with the inflow state as labels -- the arithmetic per
iteration is the reference's, the flow it converges to is not validated.

    python examples/cylinder2d_unsteady.py epochs=200
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402


def synthetic_points(cfg, rng):
    def channel(n):
        pts = np.empty((0, 2), np.float32)
        while len(pts) < n:
            p = rng.uniform([-8, -8], [25, 8], (2 * n, 2)).astype(np.float32)
            pts = np.concatenate([pts, p[np.hypot(p[:, 0], p[:, 1]) > 0.5]])
        return pts[:n]

    dom = channel(cfg["npoint_pde"])
    n_cyl = cfg["npoint_inlet_cylinder"] // 2
    n_in = cfg["npoint_inlet_cylinder"] - n_cyl
    th = np.linspace(0, 2 * np.pi, n_cyl, endpoint=False, dtype=np.float32)
    inlet_cyl = np.concatenate([np.stack([np.full(n_in, -8.0, np.float32), np.linspace(-8, 8, n_in, dtype=np.float32)], 1),
                                np.stack([0.5 * np.cos(th), 0.5 * np.sin(th)], 1)])
    uv = np.concatenate([np.tile([[1.0, 0.0]], (n_in, 1)), np.zeros((n_cyl, 2))]).astype(np.float32)
    outlet = np.stack([np.full(cfg["npoint_outlet"], 25.0, np.float32), np.linspace(-8, 8, cfg["npoint_outlet"], dtype=np.float32)], 1)
    return dom, inlet_cyl, uv, outlet


def stamped(xy, stamps, labels=None):
    """IterableCSVDataset with `timestamps` (csv_dataset.py): every point at every time stamp, time-major."""
    t = np.repeat(np.asarray(stamps, np.float32), len(xy)).reshape(-1, 1)
    rep = lambda a: np.tile(a, (len(stamps), 1))  # noqa: E731
    inp = {"t": t, "x": rep(xy[:, 0:1]), "y": rep(xy[:, 1:2])}
    return inp, ({k: rep(v) for k, v in labels.items()} if labels else None)


DEFAULTS = dict(seed=42, output_dir="./output_cylinder2d_unsteady", epochs=200, log_freq=20, viscosity=0.02, density=1.0,
                time_start=1.0, time_end=50.0, num_timestamps=50, train_num_timestamps=30, npoint_pde=9420,
                npoint_inlet_cylinder=161, npoint_outlet=81, num_layers=5, hidden_size=50, learning_rate=1e-3)


def build(cfg):
    """model, constraints and Solver of the case (also used by bench.py's `extra` entry)."""
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    rng = np.random.default_rng(cfg["seed"])
    model = ppsci.arch.MLP(("t", "x", "y"), ("u", "v", "p"), cfg["num_layers"], cfg["hidden_size"], "tanh")
    equation = {"NavierStokes": ppsci.equation.NavierStokes(cfg["viscosity"], cfg["density"], 2, True)}
    stamps = np.linspace(cfg["time_start"], cfg["time_end"], cfg["num_timestamps"], endpoint=True).astype("float32")
    train_stamps = np.sort(np.random.choice(stamps, cfg["train_num_timestamps"]))
    t0 = np.array([cfg["time_start"]], dtype="float32")
    dom, inlet_cyl, uv, outlet = synthetic_points(cfg, rng)
    geom = {"time_rect": ppsci.geometry.TimeXGeometry(
        ppsci.geometry.TimeDomain(cfg["time_start"], cfg["time_end"], timestamps=np.concatenate((t0, train_stamps), axis=0)),
        ppsci.geometry.PointCloud({"x": dom[:, 0:1], "y": dom[:, 1:2]}, ("x", "y")))}
    ntime = len(train_stamps)
    pde = ppsci.constraint.InteriorConstraint(
        equation["NavierStokes"].equations, {"continuity": 0, "momentum_x": 0, "momentum_y": 0}, geom["time_rect"],
        {"dataset": "IterableNamedArrayDataset", "batch_size": cfg["npoint_pde"] * ntime, "iters_per_epoch": 1},
        ppsci.loss.MSELoss("mean"), name="EQ")

    def sup(name, xy, stamps_, labels, weight):
        inp, lab = stamped(xy, stamps_, labels)
        w = {k: np.full_like(v, weight) for k, v in lab.items()}
        return ppsci.constraint.SupervisedConstraint(
            {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": lab, "weight": w}},
            ppsci.loss.MSELoss("mean"), name=name)

    bc_in = sup("BC_inlet_cylinder", inlet_cyl, train_stamps, {"u": uv[:, 0:1], "v": uv[:, 1:2]}, 10.0)
    bc_out = sup("BC_outlet", outlet, train_stamps, {"p": np.zeros((len(outlet), 1), np.float32)}, 1.0)
    ic = sup("IC", dom, t0, {"u": np.ones((len(dom), 1), np.float32), "v": np.zeros((len(dom), 1), np.float32),
                             "p": np.zeros((len(dom), 1), np.float32)}, 10.0)
    constraint = {c.name: c for c in (pde, bc_in, bc_out, ic)}
    optimizer = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, constraint, cfg["output_dir"], optimizer, None, cfg["epochs"], 1, log_freq=cfg["log_freq"],
                                 equation=equation, geom=geom)
    logger.info(f"points per iteration: EQ {cfg['npoint_pde'] * ntime} + BC {len(inlet_cyl) * ntime} + {len(outlet) * ntime} + IC {len(dom)}"
                f" (reference TIPC ips for this case: 1 264 165.6)")
    return solver


if __name__ == "__main__":
    cfg = parse(dict(DEFAULTS))
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    build(cfg).train()
