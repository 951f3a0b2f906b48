"""Data parallelism: one process per rank, rank-strided shards of each batch, ONE all-reduce
(sum) of the flat gradient per step, global-batch loss normalisation -- so a W-rank run reproduces the
1-rank run on the same global point set.  Runs here on CPU: gloo backend, world_size 2, kernels under the
CPU SIMT emulator.  (The reference has no distributed test at all)"""
import os
import subprocess
import sys

import numpy as np
import pytest


def _free_port() -> str:
    """A loopback port nobody listens on right now (the suite runs on several pytest-xdist workers: fixed ports collide)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(outdir, world, reduction):
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, outdir, reduction]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", _free_port(), worker, outdir, reduction]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600, stdout=subprocess.DEVNULL)
    return np.load(os.path.join(outdir, f"result_w{world}.npz"))


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_two_ranks_reproduce_single_rank(tmp_path, reduction):
    d = str(tmp_path)
    one = _run(d, 1, reduction)
    two = _run(d, 2, reduction)
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-5, atol=1e-6)
    # the logged loss of rank 0 covers its own shard only for "sum"; for "mean" it is the local part of the global mean
    assert np.isfinite(two["loss"])


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_ragged_shards_get_zero_weight_padding(tmp_path, reduction):
    """67 samples on 2 ranks: the sampler pads rank 1's shard by wrapping around (as the reference's
    DistributedBatchSampler does, /root/reference/ppsci/data/__init__.py:76-99); the duplicate gets zero weight and "mean"
    divides by 67, not 68 -- so the 2-rank run still reproduces the 1-rank run."""
    d = str(tmp_path)
    one = _run(d, 1, "ragged_" + reduction)
    two = _run(d, 2, "ragged_" + reduction)
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-5, atol=1e-6)


def test_two_ranks_reproduce_single_rank_with_batch_reductions(tmp_path):
    """`u.mean()` / `.sum()` inside the residual (three launches of the residual program, engine._forward_reductions): the sums and
    their adjoints are all-reduced, so two ranks train what one rank trains."""
    d = str(tmp_path)
    one = _run(d, 1, "batchmean")
    two = _run(d, 2, "batchmean")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-5, atol=1e-6)


def test_two_ranks_reproduce_single_rank_piratenet(tmp_path):
    """PirateNet (layer-by-layer kernels, RWF, trainable Fourier kernel and alpha): the flat trainable-layout gradient is
    all-reduced like an MLP's."""
    d = str(tmp_path)
    one = _run(d, 1, "pirate")
    two = _run(d, 2, "pirate")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-4, atol=1e-5)


def test_two_ranks_reproduce_single_rank_fno(tmp_path):
    """Operator path (TFNO2dNet): per-rank mean loss over its half of the batch, gradients averaged over ranks
    == the single-rank run on the whole batch."""
    d = str(tmp_path)
    one = _run(d, 1, "fno")
    two = _run(d, 2, "fno")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-4, atol=1e-5)


def test_two_ranks_reproduce_single_rank_uno(tmp_path):
    """The same for UNONet (uno_engine.UnoNative: resolution-changing blocks, U skip) -- the executor is a sibling of the FNO one and
    shares its data-parallel contract."""
    d = str(tmp_path)
    one = _run(d, 1, "uno")
    two = _run(d, 2, "uno")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-4, atol=1e-5)


def test_two_ranks_reproduce_single_rank_sfno(tmp_path):
    """... and for SFNONet (the FNO executor with the spherical-harmonic transform pair)."""
    d = str(tmp_path)
    one = _run(d, 1, "sfno")
    two = _run(d, 2, "sfno")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-4, atol=1e-5)


def test_two_ranks_reproduce_single_rank_spinn(tmp_path):
    """Separable path (SPINN / Helmholtz3D): x-axis slabs per rank, global-grid loss normalisation, SUM all-reduce."""
    d = str(tmp_path)
    one = _run(d, 1, "spinn")
    two = _run(d, 2, "spinn")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-4, atol=1e-5)


def test_two_ranks_reproduce_single_rank_factored_layers_and_equation_parameters(tmp_path):
    """MLP(random_weight=...) + Vibration's learnable k1, k2: network parameters (trainable layout), the two exponents
    and the prediction after three Adam steps are the same on one and on two ranks."""
    d = str(tmp_path)
    one = _run(d, 1, "viv")
    two = _run(d, 2, "viv")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-4, atol=1e-5)


def test_two_ranks_reproduce_single_rank_periodic_constraint_and_eval_gather(tmp_path):
    """PeriodicConstraint under data parallelism (each rank pairs the halves of its rank-strided shard) and the
    validator metric over a dataset whose size is not a multiple of the world size (the gathered shards are put back
    into dataset order and trimmed to len(dataset), eval.py:154-161)."""
    d = str(tmp_path)
    one = _run(d, 1, "periodic")
    two = _run(d, 2, "periodic")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-5)


def test_iterable_dataset_refuses_world_size_gt_1(tmp_path):
    """data/__init__.py:62-66: refused in a real two-rank process group (every rank), accepted on one rank."""
    import json

    import ppsci.data as D

    class FakeDS:
        is_iterable = True

    import torch.distributed as dist

    assert not dist.is_initialized()
    assert D.build_dataloader(FakeDS(), {}) is not None
    d = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", _free_port(), worker, d, "iterable"], check=True, env=env, cwd=ROOT, timeout=300,
                   stdout=subprocess.DEVNULL)
    for rank in (0, 1):
        r = json.load(open(os.path.join(d, f"iterable_w2_r{rank}.json")))
        assert r["world"] == 2 and r["message"] == "world_size(2) should be 1 when using IterableDataset." and r["extension"]


def test_ranks_must_agree_on_trace_time_branches(tmp_path):
    """graph.batch_values follows Python control flow on the values of a fixed batch; two ranks whose shards answer differently
    are refused on both ranks with the two answers in the message, a shard-independent condition trains."""
    import json

    d = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", _free_port(), worker, d, "branch"], check=True, env=env, cwd=ROOT, timeout=600, stdout=subprocess.DEVNULL)
    for rank in (0, 1):
        r = json.load(open(os.path.join(d, f"branch_w2_r{rank}.json")))
        assert r["shard_independent"] == "trained"
        assert "ranks 0 and 1 took different branches" in r["shard_dependent"] and "float(x[0:1]) = 0.5" in r["shard_dependent"] \
            and "float(x[0:1]) = -0.5" in r["shard_dependent"]


def test_two_ranks_reproduce_single_rank_fused_tile_kernel(tmp_path):
    """Padded width 64: under data parallelism the step keeps the fused tile kernel -- tile kernel + tail kernel (sums) ->
    ONE all-reduce of the flat gradient -> Adam and the next step's weight fragments in one launch
    (ppsci_taylor_step_plan_apply) -- and reproduces the single-rank run (whose Adam sits in the tail kernel)."""
    d = str(tmp_path)
    one = _run(d, 1, "fused64")
    two = _run(d, 2, "fused64")
    np.testing.assert_allclose(two["params"], one["params"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(two["pred"], one["pred"], rtol=1e-5, atol=1e-6)


def test_batch_sampler_rank_strided_shards():
    from ppsci.data import BatchSampler

    a = list(BatchSampler(10, 3, world=2, rank=0))
    b = list(BatchSampler(10, 3, world=2, rank=1))
    assert np.concatenate(a).tolist() == [0, 2, 4, 6, 8]
    assert np.concatenate(b).tolist() == [1, 3, 5, 7, 9]
    c = list(BatchSampler(7, 2, world=2, rank=1))  # padded by wrapping around to a multiple of world
    assert np.concatenate(c).tolist() == [1, 3, 5, 0]
