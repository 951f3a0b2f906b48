"""Worker of tests/test_distributed.py: one rank of a gloo data-parallel run on the CPU SIMT emulator."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_solver(outdir, world_batch, reduction, steps, pirate=False, ragged=False, width=16):
    import ppsci
    from oracle import taylor_np as T
    from tests.common import set_model_weights

    if pirate:  # layer-by-layer PirateNet path: period + Fourier embedding, RWF, gates, alpha
        ppsci.utils.misc.set_random_seed(5)
        model = ppsci.arch.PirateNet(("t", "x"), ("u",), 2, 16, "tanh", periods={"x": (2.0, False)},
                                     fourier={"dim": 16, "scale": 1.0}, random_weight={"mean": 1.0, "std": 0.1})
        with torch.no_grad():
            for n_, v_ in model.named_parameters():
                if n_.endswith("alpha"):
                    v_.fill_(0.4)
    else:
        model = ppsci.arch.MLP(("t", "x"), ("u",), 2, width, "tanh")
        set_model_weights(model, T.make_net(2, [width, width], 1, seed=7, bias_scale=0.05))
    N = world_batch
    X = np.random.default_rng(3).uniform([0, -1], [1, 1], (N, 2)).astype(np.float32)
    lab = np.random.default_rng(4).standard_normal((N, 1)).astype(np.float32) * 0.1
    eq = ppsci.equation.AllenCahn(eps=0.01)
    world = dist.get_world_size() if dist.is_initialized() else 1
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"t": X[:, :1], "x": X[:, 1:]},
                       "label": {"allen_cahn": lab}},
           "batch_size": N // world, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    if ragged:  # N % world != 0: one batch per epoch of ceil(N / world) samples per rank, the last rank's padded
        cfg["batch_size"] = -(-N // world)
        cfg["sampler"]["drop_last"] = False
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(reduction), eq.equations, name="EQ")
    opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    return ppsci.solver.Solver(model, {"EQ": cst}, outdir, opt, epochs=steps, iters_per_epoch=1, log_freq=1), model


def build_reduction_solver(outdir, world_batch, steps):
    """A residual with batch reductions (graph.Sym.mean / .sum): under data parallelism the sums and their adjoints are
    all-reduced between the launches, so that the mean is over the GLOBAL batch as in a one-rank run."""
    import ppsci
    from oracle import taylor_np as T
    from ppsci.autodiff import jacobian
    from tests.common import set_model_weights

    model = ppsci.arch.MLP(("t", "x"), ("u",), 2, 16, "tanh")
    set_model_weights(model, T.make_net(2, [16, 16], 1, seed=7, bias_scale=0.05))
    N = world_batch
    X = np.random.default_rng(3).uniform([0, -1], [1, 1], (N, 2)).astype(np.float32)
    lab = np.random.default_rng(4).standard_normal((N, 1)).astype(np.float32) * 0.1
    world = dist.get_world_size() if dist.is_initialized() else 1
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"t": X[:, :1], "x": X[:, 1:]}, "label": {"r": lab}},
           "batch_size": N // world, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    # (no exact invariance in it: `u - u.mean()` would leave the last bias with a zero gradient, which Adam turns into +-lr noise)
    exprs = {"r": lambda d: d["u"] * (1.0 + (d["u"] * d["u"]).mean()) + (jacobian(d["u"], d["x"]) * d["t"]).sum() / 50.0}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), exprs, name="EQ")
    opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    return ppsci.solver.Solver(model, {"EQ": cst}, outdir, opt, epochs=steps, iters_per_epoch=1, log_freq=1), model


def build_viv_solver(outdir, world_batch, steps):
    """Factored layers (random_weight) + learnable equation parameters (Vibration): the kernel-layout gradient is
    all-reduced and pulled back, the equation-parameter gradient has its own all-reduce."""
    import ppsci
    from paddlescience_amd.equation.pde.base import EqParamStore

    EqParamStore.reset()
    ppsci.utils.misc.set_random_seed(11)
    model = ppsci.arch.MLP(("t_f",), ("eta",), 2, 16, "tanh", random_weight={"mean": 0.5, "std": 0.1})
    eq = ppsci.equation.Vibration(1.5, 0.3, -0.2)
    N = world_batch
    rng = np.random.default_rng(3)
    t = rng.uniform(0, 1, (N, 1)).astype(np.float32)
    lab = {"eta": rng.standard_normal((N, 1)).astype(np.float32) * 0.1, "f": rng.standard_normal((N, 1)).astype(np.float32)}
    world = dist.get_world_size() if dist.is_initialized() else 1
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"t_f": t}, "label": lab},
           "batch_size": N // world, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"),
                                                {"eta": lambda out: out["eta"], **eq.equations}, name="Sup")
    opt = ppsci.optimizer.Adam(learning_rate=1e-2)((model, eq))
    solver = ppsci.solver.Solver(model, {"Sup": cst}, outdir, opt, epochs=steps, iters_per_epoch=1, log_freq=1,
                                 equation={"VIV": eq})
    return solver, model, eq


def main_viv(outdir):
    from paddlescience_amd import device
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    solver, model, eq = build_viv_solver(outdir, 48, 3)
    solver.train()
    pred = solver.predict({"t_f": np.linspace(0, 1, 9, dtype=np.float32).reshape(-1, 1)}, batch_size=4, return_numpy=True)
    if not dist.is_initialized() or dist.get_rank() == 0:
        np.savez(os.path.join(outdir, f"result_w{world}.npz"),
                 params=np.concatenate([model.flat_params.numpy(), [eq.k1.item(), eq.k2.item()]]), pred=pred["eta"],
                 loss=np.asarray(solver.last_losses["loss"]))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def build_fno_solver(outdir, world_batch, steps, uno=False, sfno=False):
    """Operator-learning path: TFNO2dNet (or UNONet) on the native executor, gradient averaged over ranks (DataParallel
    semantics)."""
    import ppsci

    torch.manual_seed(5)
    if uno:
        model = ppsci.arch.UNONet(("x",), ("y",), 3, 1, 6, lifting_channels=8, projection_channels=8, n_layers=3,
                                  uno_out_channels=[4, 6, 4], uno_n_modes=[[8, 8], [4, 4], [4, 4]],
                                  uno_scalings=[[0.5, 0.5], [1, 1], [2, 2]], norm="group_norm")
    elif sfno:
        model = ppsci.arch.SFNONet(("x",), ("y",), (8, 8), 6, in_channels=3, out_channels=1, lifting_channels=8, projection_channels=8,
                                   n_layers=2, norm="group_norm")
    else:
        model = ppsci.arch.TFNO2dNet(("x",), ("y",), 4, 4, hidden_channels=8, lifting_channels=16, projection_channels=16,
                                     n_layers=2, norm="group_norm")
    rng = np.random.default_rng(9)
    x = rng.standard_normal((world_batch, 3, 8, 8)).astype(np.float32)
    y = rng.standard_normal((world_batch, 1, 8, 8)).astype(np.float32)
    world = dist.get_world_size() if dist.is_initialized() else 1

    def mse(output_dict, label_dict, weight_dict=None):
        return {"l2": ((output_dict["y"] - label_dict["y"]) ** 2).mean()}

    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": x}, "label": {"y": y}},
           "batch_size": world_batch // world, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.FunctionalLoss(mse), name="Sup")
    opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    return ppsci.solver.Solver(model, {"Sup": cst}, outdir, opt, epochs=steps, iters_per_epoch=1, log_freq=1), model, x


def main():
    outdir, reduction = sys.argv[1], sys.argv[2]
    if reduction in ("fno", "uno", "sfno"):
        return main_fno(outdir, reduction == "uno", reduction == "sfno")
    if reduction == "spinn":
        return main_spinn(outdir)
    if reduction == "viv":
        return main_viv(outdir)
    if reduction == "periodic":
        return main_periodic(outdir)
    if reduction == "iterable":
        return main_iterable(outdir)
    if reduction == "branch":
        return main_branch(outdir)
    from paddlescience_amd import device
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    pirate = reduction == "pirate"
    ragged = reduction.startswith("ragged_")
    if reduction == "fused64":
        # padded width 64: the fused tile kernel; under data parallelism its step is tile kernel + tail kernel -> all-reduce ->
        # Adam + fragments in one launch (engine.Engine.train_step, ppsci_taylor_step_plan_apply); four steps, so that the
        # fragments the apply kernel left behind are used (the weight split is skipped from the second step on)
        solver, model = build_solver(outdir, 64, "mean", 4, width=64)
    elif reduction == "batchmean":
        solver, model = build_reduction_solver(outdir, 64, 3)
    elif ragged:
        solver, model = build_solver(outdir, 67, reduction[len("ragged_"):], 2, ragged=True)
    else:
        solver, model = build_solver(outdir, 64, "mean" if pirate else reduction, 2, pirate=pirate)
    solver.train()
    pred = solver.predict({"t": np.linspace(0, 1, 11, dtype=np.float32).reshape(-1, 1),
                           "x": np.linspace(-1, 1, 11, dtype=np.float32).reshape(-1, 1)}, batch_size=4, return_numpy=True)
    if not dist.is_initialized() or dist.get_rank() == 0:
        np.savez(os.path.join(outdir, f"result_w{world}.npz"), params=model.flat_params.numpy(), pred=pred["u"],
                 loss=np.asarray(solver.last_losses["loss"]))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_iterable(outdir):
    """data/__init__.py:62-66 of the reference: an IterableDataset under world_size > 1 is refused -- in a REAL two-rank
    process group."""
    import ppsci.data as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")

    class FakeDS:
        is_iterable = True

    msg = ""
    try:
        D.build_dataloader(FakeDS(), {})
    except ValueError as e:
        msg = str(e)
    ok_ext = D.build_dataloader(FakeDS(), {"shard_in_engine": True}) is not None  # (this framework's SPINN extension)
    rank = dist.get_rank() if dist.is_initialized() else 0
    with open(os.path.join(outdir, f"iterable_w{world}_r{rank}.json"), "w") as f:
        json.dump({"world": dist.get_world_size() if dist.is_initialized() else 1, "message": msg, "extension": ok_ext}, f)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_branch(outdir):
    """Python control flow on the values of a fixed batch, under two ranks whose shards give DIFFERENT answers: refused on every
    rank (solver.Solver._check_trace_decisions); a condition every shard answers alike is accepted."""
    import ppsci
    from paddlescience_amd import device
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    rank = dist.get_rank() if dist.is_initialized() else 0
    N = 32
    X = np.random.default_rng(3).uniform([0, -1], [1, 1], (N, 2)).astype(np.float32)
    X[0, 1], X[1, 1] = 0.5, -0.5  # row 0 goes to rank 0, row 1 to rank 1 (rank-strided shards)
    out = {}
    for tag, cond in (("shard_dependent", lambda d: float(d["x"][0]) > 0.0), ("shard_independent", lambda d: float(d["t"][0]) >= 0.0)):
        model = ppsci.arch.MLP(("t", "x"), ("u",), 2, 16, "tanh")
        cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"t": X[:, :1], "x": X[:, 1:]}, "label": {"v": np.zeros((N, 1), np.float32)}},
               "batch_size": N // world, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
        cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"),
                                                    {"v": lambda d, cond=cond: d["u"] * (2.0 if cond(d) else 1.0)}, name="EQ")
        opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
        try:
            ppsci.solver.Solver(model, {"EQ": cst}, os.path.join(outdir, tag), opt, epochs=1, iters_per_epoch=1).train()
            out[tag] = "trained"
        except RuntimeError as e:
            out[tag] = str(e)
    with open(os.path.join(outdir, f"branch_w{world}_r{rank}.json"), "w") as f:
        json.dump(out, f)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_periodic(outdir):
    """PeriodicConstraint + PeriodicMSELoss next to a supervised constraint with a validator: the batch
    [boundary points ; their images] is sharded rank-strided, so every rank pairs the halves of its own shard
    (needs an even per-rank half); eval gathers the ranks' shards back into dataset order without the sampler's
    wrap-around padding (9 validation samples on 2 ranks)."""
    import sympy as sp

    import ppsci
    from oracle import taylor_np as T
    from paddlescience_amd import device
    from tests.common import set_model_weights
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    geom = ppsci.geometry.Rectangle((-1.0, 0.0), (1.0, 2.0))
    model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16, "tanh")
    set_model_weights(model, T.make_net(2, [16, 16], 1, seed=7, bias_scale=0.05))
    x, y = sp.symbols("x y")
    u = sp.Function("u")(x, y)
    np.random.seed(7)
    pbc = ppsci.constraint.PeriodicConstraint(
        {"u": lambda out: out["u"], "u_x": u.diff(x)}, {"u": 0, "u_x": 0}, geom, "x",
        {"dataset": "NamedArrayDataset", "batch_size": 16, "iters_per_epoch": 1,
         "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": False}},
        ppsci.loss.PeriodicMSELoss("mean", weight={"u_x": 0.5}), criteria=lambda x, y: np.isclose(x, -1.0), name="PBC")
    rng = np.random.default_rng(5)
    Xv = rng.uniform([-1, 0], [1, 2], (9, 2)).astype(np.float32)
    lv = rng.standard_normal((9, 1)).astype(np.float32)
    val = ppsci.validate.SupervisedValidator(
        {"dataset": {"name": "NamedArrayDataset", "input": {"x": Xv[:, :1], "y": Xv[:, 1:]}, "label": {"u": lv}},
         "batch_size": 5, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": False}},
        ppsci.loss.MSELoss("mean"), {"u": lambda out: out["u"]}, metric={"MSE": ppsci.metric.MSE()}, name="V")
    opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PBC": pbc}, outdir, opt, epochs=3, iters_per_epoch=1, log_freq=1,
                                 validator={"V": val})
    solver.train()
    target, group = solver.eval()
    if not dist.is_initialized() or dist.get_rank() == 0:
        np.savez(os.path.join(outdir, f"result_w{world}.npz"), params=model.flat_params.numpy(),
                 pred=np.asarray([target]), loss=np.asarray(solver.last_losses["loss"]))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_spinn(outdir):
    """Separable path: the x-axis points are rank-strided (each rank owns an [nx/W, ny, nz] slab), the loss is
    normalised by the global grid size, gradients are summed."""
    import ppsci
    from paddlescience_amd import device
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    np.random.seed(11)
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), r=4, num_layers=2, hidden_size=16, activation="tanh")
    eq = ppsci.equation.Helmholtz(3, 1.0)
    eq.model = model
    rng = np.random.default_rng(3)
    shape = (6, 5, 4)
    xs = [rng.uniform(-1, 1, (n, 1)).astype(np.float32) for n in shape]
    uc = rng.standard_normal(shape + (1,)).astype(np.float32)
    data = {"x": xs[0], "y": xs[1], "z": xs[2], "uc": uc}
    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: data, "label": lambda d: {"helmholtz": d["uc"]}},
         "shard_in_engine": True},
        output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
    face = {"x": np.asarray([[1.0]], np.float32), "y": xs[1], "z": xs[2]}  # x-axis of one point: sharded along y
    bc = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: face,
                     "label": lambda d: {"u": np.zeros([1, shape[1], shape[2], 1], np.float32)}}, "shard_in_engine": True},
        output_expr={"u": lambda out: out["u"]}, loss=ppsci.loss.MSELoss("mean"), name="BC0")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PDE": pde, "BC0": bc}, outdir, opt, epochs=2, iters_per_epoch=1, log_freq=1,
                                 equation={"Helmholtz": eq})
    solver.train()
    pred = solver.predict({"x": xs[0], "y": xs[1], "z": xs[2]}, batch_size=None, return_numpy=True)
    if not dist.is_initialized() or dist.get_rank() == 0:
        np.savez(os.path.join(outdir, f"result_w{world}.npz"), params=model.flat_params.numpy(), pred=pred["u"],
                 loss=np.asarray(solver.last_losses["loss"]))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_fno(outdir, uno=False, sfno=False):
    from paddlescience_amd import device
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    solver, model, x = build_fno_solver(outdir, 4, 2, uno, sfno)
    solver.train()
    pred = solver.predict({"x": x}, return_numpy=True)
    if not dist.is_initialized() or dist.get_rank() == 0:
        np.savez(os.path.join(outdir, f"result_w{world}.npz"), params=model.flat_params.numpy(), pred=pred["y"],
                 loss=np.asarray(solver.last_losses["loss"]))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
