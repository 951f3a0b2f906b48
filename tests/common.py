"""Shared test plumbing: the `dev` fixture logic (CPU SIMT emulator vs real MI355X)."""
import numpy as np
import pytest
import torch


def make_dev_fixture():
    @pytest.fixture(autouse=True, params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
    def dev(request):
        from paddlescience_amd import _lib, device
        from tests.emu import build_emu

        if request.param == "emu":
            build_emu.inject()
            device.set_device("cpu")
        else:
            _lib._inject_for_tests(None)
            device.set_device(None)
        yield request.param
        _lib._inject_for_tests(None)
        device.set_device(None)

    return dev


def set_model_weights(model, net):
    """Copy oracle NetSpec weights into a paddlescience_amd MLP."""
    from oracle import taylor_np as T

    flat = torch.tensor(T.flat_params(net), dtype=torch.float32)
    assert flat.numel() == model.flat_params.numel()
    model.flat_params.copy_(flat.to(model.flat_params.device))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
