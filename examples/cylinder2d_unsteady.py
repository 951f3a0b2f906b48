"""2-D unsteady flow around a cylinder, Re = 100 -- /root/reference/examples/cylinder/2d_unsteady/cylinder2d_unsteady_Re100.py
(+ conf/cylinder2d_unsteady_Re100.yaml): MLP (t, x, y) -> (u, v, p), 5 x 50 tanh; NavierStokes(nu 0.02, rho 1, dim 2, time);
per iteration ONE full batch of every constraint: 9 420 domain points x 30 time stamps for the PDE, 161 inlet + cylinder
points and 81 outlet points x 30 time stamps, 9 420 initial-condition points, weights 10 on the velocity labels.  This is the
case behind the reference's published TIPC throughput (`ips` 1 264 165.6 points/s, fp32, one unnamed NVIDIA GPU,
test_tipc/README.MD:17): `ips` in the [Train] log lines below is the same quantity (train.py:106, printer.py:66).

The reference reads the point sets and OpenFOAM labels from ./datasets/*.csv (download_dataset.py; no network here) through
`reader.load_csv_file` / `IterableCSVDataset`, and so does this script: when `data_dir` does not hold the tables, CSV files of the same
sizes and column names are written first (uniform points of the channel [-8, 25] x [-8, 8] minus
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


def ensure_data(cfg, rng):
    """The four CSV tables of the reference's dataset (domain_train / domain_inlet_cylinder / domain_outlet / initial/ic0.1) in
    its column naming (`Points:0`, `Points:1`, `U:0`, `U:1`, `p`); written with synthetic content when missing."""
    d = cfg["data_dir"]
    paths = {k: os.path.join(d, f) for k, f in (("dom", "domain_train.csv"), ("in", "domain_inlet_cylinder.csv"),
                                                 ("out", "domain_outlet.csv"), ("ic", os.path.join("initial", "ic0.1.csv")))}
    if all(os.path.exists(f) for f in paths.values()):
        return paths
    logger.warning(f"{d}: CSV tables not found, writing synthetic stand-ins of the reference's sizes")
    os.makedirs(os.path.join(d, "initial"), exist_ok=True)
    dom, inlet_cyl, uv, outlet = synthetic_points(cfg, rng)

    def write(path, cols):
        keys = list(cols)
        np.savetxt(path, np.stack([cols[k] for k in keys], 1), delimiter=",", header=",".join(keys), comments="", fmt="%.8g")

    write(paths["dom"], {"Points:0": dom[:, 0], "Points:1": dom[:, 1]})
    write(paths["in"], {"Points:0": inlet_cyl[:, 0], "Points:1": inlet_cyl[:, 1], "U:0": uv[:, 0], "U:1": uv[:, 1]})
    write(paths["out"], {"Points:0": outlet[:, 0], "Points:1": outlet[:, 1], "p": np.zeros(len(outlet))})
    write(paths["ic"], {"Points:0": dom[:, 0], "Points:1": dom[:, 1], "U:0": np.ones(len(dom)), "U:1": np.zeros(len(dom)),
                        "p": np.zeros(len(dom))})
    return paths


DEFAULTS = dict(seed=42, output_dir="./output_cylinder2d_unsteady", data_dir="./datasets/cylinder2d_unsteady", epochs=200, log_freq=20, viscosity=0.02, density=1.0,
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
    paths = ensure_data(cfg, rng)
    alias = {"x": "Points:0", "y": "Points:1", "u": "U:0", "v": "U:1"}
    geom = {"time_rect": ppsci.geometry.TimeXGeometry(
        ppsci.geometry.TimeDomain(cfg["time_start"], cfg["time_end"], timestamps=np.concatenate((t0, train_stamps), axis=0)),
        ppsci.geometry.PointCloud(ppsci.utils.reader.load_csv_file(paths["dom"], ("x", "y"), alias), ("x", "y")))}
    ntime = len(train_stamps)
    pde = ppsci.constraint.InteriorConstraint(
        equation["NavierStokes"].equations, {"continuity": 0, "momentum_x": 0, "momentum_y": 0}, geom["time_rect"],
        {"dataset": "IterableNamedArrayDataset", "batch_size": cfg["npoint_pde"] * ntime, "iters_per_epoch": 1},
        ppsci.loss.MSELoss("mean"), name="EQ")

    def sup(name, path, labels, stamps_, weight):  # cylinder2d_unsteady_Re100.py:98-155
        ds = {"name": "IterableCSVDataset", "file_path": path, "input_keys": ("x", "y"), "label_keys": labels,
              "alias_dict": alias, "timestamps": stamps_}
        if weight:
            ds["weight_dict"] = {k: weight for k in labels}
        return ppsci.constraint.SupervisedConstraint({"dataset": ds}, ppsci.loss.MSELoss("mean"), name=name)

    bc_in = sup("BC_inlet_cylinder", paths["in"], ("u", "v"), train_stamps, 10)
    bc_out = sup("BC_outlet", paths["out"], ("p",), train_stamps, None)
    ic = sup("IC", paths["ic"], ("u", "v", "p"), t0, 10)
    constraint = {c.name: c for c in (pde, bc_in, bc_out, ic)}
    optimizer = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, constraint, cfg["output_dir"], optimizer, None, cfg["epochs"], 1, log_freq=cfg["log_freq"],
                                 equation=equation, geom=geom)
    logger.info("points per iteration: " + " + ".join(f"{n} {getattr(c.data_loader, 'dataset', c.data_loader).num_samples}" for n, c in constraint.items())
                + " (reference TIPC ips for this case: 1 264 165.6)")
    return solver


if __name__ == "__main__":
    cfg = parse(dict(DEFAULTS))
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    build(cfg).train()
