"""Every script under examples/ runs end to end through the ppsci API at a tiny size (CPU SIMT emulator): catches drift between
the examples and the package.  The GPU halves are the end-to-end runs quoted in DESIGN.md section 5."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TINY = {
    "laplace2d.py": ["epochs=3", "npoint_interior=64", "npoint_bc=16", "log_freq=1"],
    "allen_cahn_plain.py": ["epochs=1", "iters_per_epoch=2", "batch_size=32", "hidden_size=32", "num_layers=2", "log_freq=1", "n_x=32", "n_t_eval=5"],
    "allen_cahn_piratenet.py": ["epochs=1", "iters_per_epoch=1", "batch_size=32", "hidden_size=16", "num_blocks=1", "log_freq=1",
                                "grad_norm_update_freq=1", "n_x=32", "n_t_eval=5"],
    "euler_beam.py": ["epochs=3", "log_freq=1"],
    "cylinder2d_unsteady.py": ["epochs=2", "npoint_pde=40", "npoint_inlet_cylinder=11", "npoint_outlet=5", "train_num_timestamps=3",
                               "num_timestamps=5", "log_freq=1"],
    "ldc2d_steady.py": ["epochs=1", "iters_per_epoch=2", "num_layers=2", "hidden_size=16", "npoint_pde=49", "npoint_bc=16", "log_freq=1"],
    "poiseuille_flow.py": ["epochs=1", "batch_size=16", "N_x=3", "N_y=4", "N_p=4", "log_freq=1"],
    "spinn_helmholtz3d.py": ["epochs=1", "iters_per_epoch=2", "nc=6", "nc_test=5", "r=4", "num_layers=2", "hidden_size=8", "log_freq=1",
                             "resample_every=1"],
    "tfno_darcyflow.py": ["epochs=1", "n_train=4", "n_test=2", "batch_size=2", "n_modes=4", "hidden_channels=8", "lifting_channels=8",
                          "projection_channels=8", "n_layers=1", "log_freq=1"],
    "sfno_swe.py": ["epochs=1", "n_train=4", "n_test=2", "batch_size=2", "n_modes=8", "hidden_channels=6", "lifting_channels=8",
                    "projection_channels=8", "n_layers=2", "log_freq=1"],
    "uno_darcyflow.py": ["epochs=1", "n_train=4", "n_test=2", "batch_size=2", "width=0.125", "modes=0.5", "lifting_channels=8",
                         "projection_channels=8", "log_freq=1"],
    "ldc_2d_sota.py": ["epochs=1,1", "Re=100,400", "iters_per_epoch=1", "hidden_size=16", "num_layers=2", "fourier_dim=8", "batch_pde=32",
                       "batch_bc=8", "log_freq=1", "grad_norm_update_freq=1", "n_eval=9"],
}


@pytest.mark.parametrize("script", sorted(TINY))
def test_example_runs(script, tmp_path, monkeypatch):
    from paddlescience_amd import _lib, device
    from tests.emu import build_emu

    build_emu.inject()
    device.set_device("cpu")
    args = TINY[script] + [f"output_dir={tmp_path}/out"]
    if script in ("cylinder2d_unsteady.py", "tfno_darcyflow.py", "uno_darcyflow.py", "sfno_swe.py"):
        args.append(f"data_dir={tmp_path}/data")
    monkeypatch.setattr(sys, "argv", [script] + args)
    try:
        runpy.run_path(os.path.join(ROOT, "examples", script), run_name="__main__")
    finally:
        _lib._inject_for_tests(None)
        device.set_device(None)
