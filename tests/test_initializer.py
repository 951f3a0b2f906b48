"""ppsci.utils.initializer: the distributions / fans / gains of /root/reference/ppsci/utils/initializer.py, on torch tensors
(also on views of a model's flat parameter buffer)."""
import math

import numpy as np
import pytest
import torch

import ppsci
from ppsci.utils import initializer as I


def test_module_is_public_under_both_names():
    import paddlescience_amd.utils.initializer as native

    assert ppsci.utils.initializer is native
    assert set(I.__all__) == {"uniform_", "normal_", "trunc_normal_", "glorot_normal_", "constant_", "ones_", "zeros_",
                              "xavier_uniform_", "xavier_normal_", "kaiming_uniform_", "kaiming_normal_", "linear_init_",
                              "conv_init_"}


def test_fans_follow_the_reverse_flag():
    w = torch.empty(3, 5)
    assert I._calculate_fan_in_and_fan_out(w) == (5, 3)            # [fout, fin]
    assert I._calculate_fan_in_and_fan_out(w, reverse=True) == (3, 5)  # linear weight [fin, fout]
    conv = torch.empty(8, 4, 3, 3)
    assert I._calculate_fan_in_and_fan_out(conv) == (36, 72)
    with pytest.raises(ValueError):
        I._calculate_fan_in_and_fan_out(torch.empty(4))


def test_gains():
    assert I._calculate_gain("tanh") == 5.0 / 3 and I._calculate_gain("relu") == math.sqrt(2.0)
    assert I._calculate_gain("leaky_relu", 0.2) == math.sqrt(2.0 / 1.04) and I._calculate_gain("selu") == 0.75
    assert I._calculate_gain("sigmoid") == 1 and I._calculate_gain("conv2d") == 1
    with pytest.raises(ValueError):
        I._calculate_gain("gelu")


def test_distributions_have_the_stated_moments():
    np.random.seed(0)
    w = torch.empty(400, 300)
    I.xavier_uniform_(w, reverse=True)
    k = math.sqrt(3.0) * math.sqrt(2.0 / 700)
    assert float(w.abs().max()) <= k and abs(float(w.std()) - k / math.sqrt(3.0)) < 0.02 * k
    I.kaiming_normal_(w, reverse=True)  # examples/aneurysm/aneurysm_flow.py:108: fan_in = 400, gain sqrt(2 / (1 + 0^2))
    assert abs(float(w.std()) - math.sqrt(2.0) / math.sqrt(400)) < 2e-3
    I.trunc_normal_(w, 0.0, 0.5, -0.6, 0.7)  # examples/xpinn/model.py:106
    assert float(w.min()) >= -0.6 and float(w.max()) <= 0.7 and abs(float(w.mean())) < 0.03
    I.glorot_normal_(w)
    # a standard normal truncated to +-2 has standard deviation 0.8796...; the reference scales by sqrt(2/(fin+fout)) x 0.8796...
    expect = 0.87962566103423978 ** 2 * math.sqrt(2.0 / 700)
    assert abs(float(w.std()) - expect) < 0.02 * expect
    assert float(w.abs().max()) <= 2 * 0.87962566103423978 * math.sqrt(2.0 / 700) + 1e-7
    I.constant_(w, 0.05)
    assert float(w.min()) == float(w.max()) == pytest.approx(0.05)
    assert float(I.ones_(w).sum()) == w.numel() and float(I.zeros_(w).abs().sum()) == 0.0


def test_seeded_draws_repeat_and_write_through_parameter_views():
    from paddlescience_amd import device

    device.set_device("cpu")
    try:
        model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16)
        weights = [p for p in model.parameters() if p.ndim == 2]
        ppsci.utils.misc.set_random_seed(7)
        for p in weights:
            I.kaiming_normal_(p, reverse=True)
        first = model.flat_params.clone()
        ppsci.utils.misc.set_random_seed(7)
        for p in weights:
            I.kaiming_normal_(p, reverse=True)
        assert torch.equal(first, model.flat_params)  # same seed, same values -- and they landed in the flat buffer
        assert float(first.abs().sum()) > 0
    finally:
        device.set_device(None)


def test_linear_and_conv_defaults():
    class Lin:
        weight, bias = torch.empty(64, 32), torch.empty(32)

    np.random.seed(1)
    I.linear_init_(Lin)
    bound = 1 / math.sqrt(64)
    assert float(Lin.bias.abs().max()) <= bound
    # kaiming-uniform with a = sqrt(5) on the DEFAULT layout ([fout, fin]: fan_in = 32): bound = sqrt(3) sqrt(2/6) / sqrt(32)
    assert float(Lin.weight.abs().max()) <= math.sqrt(3.0) * math.sqrt(2.0 / 6.0) / math.sqrt(32) + 1e-7
