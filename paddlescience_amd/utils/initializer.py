"""ppsci.utils.initializer (/root/reference/ppsci/utils/initializer.py:35-498): in-place initialisers for parameter tensors.

The tensors here are torch tensors -- typically views into a model's flat parameter buffer (`model.parameters()`), on
the host or on the device.  Values are drawn on the host from numpy's global generator (the one
`ppsci.utils.misc.set_random_seed` seeds) and copied in, so a seeded script initialises identically on the CPU emulator
and on the GPU.  Distributions, fan computation (`reverse=True` for `[in, out]` linear weights) and gains are the
reference's; the random STREAM is numpy's, not paddle's (draw-for-draw equality with paddle is not attainable)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import logger

__all__ = ["uniform_", "normal_", "trunc_normal_", "glorot_normal_", "constant_", "ones_", "zeros_", "xavier_uniform_",
           "xavier_normal_", "kaiming_uniform_", "kaiming_normal_", "linear_init_", "conv_init_"]


def _assign(tensor: torch.Tensor, values: np.ndarray) -> torch.Tensor:
    with torch.no_grad():
        tensor.copy_(torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(tensor.dtype).reshape(tensor.shape))
    return tensor


def uniform_(tensor: torch.Tensor, a: float, b: float) -> torch.Tensor:
    """initializer.py:112-129: U(a, b)."""
    return _assign(tensor, np.random.uniform(a, b, size=tuple(tensor.shape)))


def normal_(tensor: torch.Tensor, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """initializer.py:132-151: N(mean, std^2)."""
    return _assign(tensor, np.random.normal(mean, std, size=tuple(tensor.shape)))


def trunc_normal_(tensor: torch.Tensor, mean: float = 0.0, std: float = 1.0, a: float = -2.0, b: float = 2.0) -> torch.Tensor:
    """initializer.py:66-103, :154-179: N(mean, std^2) restricted to [a, b] by the inverse-CDF method (a uniform draw
    between the CDF values of the bounds, mapped back through erfinv), clipped to the bounds."""
    from scipy.special import erfinv

    def cdf(x):
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0

    if mean < a - 2 * std or mean > b + 2 * std:
        logger.warning(f"mean({mean}) is more than 2 std({std}) from [a, b]([{a}, {b}]) in trunc_normal_. "
                       "The distribution of values may be incorrect.")
    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    u = np.random.uniform(2 * lo - 1, 2 * hi - 1, size=tuple(tensor.shape))
    return _assign(tensor, np.clip(erfinv(u) * (std * math.sqrt(2.0)) + mean, a, b))


def constant_(tensor: torch.Tensor, value: float = 0.0) -> torch.Tensor:
    with torch.no_grad():
        tensor.fill_(value)
    return tensor


def ones_(tensor: torch.Tensor) -> torch.Tensor:
    return constant_(tensor, 1.0)


def zeros_(tensor: torch.Tensor) -> torch.Tensor:
    return constant_(tensor, 0.0)


def _calculate_fan_in_and_fan_out(tensor: torch.Tensor, reverse: bool = False):
    """initializer.py:237-267: `reverse=False` reads the shape as [fout, fin, ...] (conv weights), `reverse=True` as
    [fin, fout] (linear weights); trailing axes are the receptive field."""
    if tensor.ndim < 2:
        raise ValueError(f"tensor.ndim should be no less than 2, but got {tensor.ndim}.")
    n_in, n_out = (tensor.shape[0], tensor.shape[1]) if reverse else (tensor.shape[1], tensor.shape[0])
    field = int(np.prod(tensor.shape[2:])) if tensor.ndim > 2 else 1
    return n_in * field, n_out * field


def xavier_uniform_(tensor: torch.Tensor, gain: float = 1.0, reverse: bool = False) -> torch.Tensor:
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor, reverse)
    k = math.sqrt(3.0) * gain * math.sqrt(2.0 / float(fan_in + fan_out))
    return uniform_(tensor, -k, k)


def xavier_normal_(tensor: torch.Tensor, gain: float = 1.0, reverse: bool = False) -> torch.Tensor:
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor, reverse)
    return normal_(tensor, 0.0, gain * math.sqrt(2.0 / float(fan_in + fan_out)))


def _calculate_correct_fan(tensor, mode: str, reverse: bool = False):
    mode = mode.lower()
    if mode not in ("fan_in", "fan_out"):
        raise ValueError(f"Mode {mode} not supported, please use one of ['fan_in', 'fan_out']")
    fan_in, fan_out = _calculate_fan_in_and_fan_out(tensor, reverse)
    return fan_in if mode == "fan_in" else fan_out


def _calculate_gain(nonlinearity: str, param=None) -> float:
    """initializer.py:333-365 (torch.nn.init.calculate_gain's table)."""
    if nonlinearity in ("linear", "conv1d", "conv2d", "conv3d", "conv_transpose1d", "conv_transpose2d", "conv_transpose3d",
                        "sigmoid"):
        return 1.0
    if nonlinearity == "tanh":
        return 5.0 / 3
    if nonlinearity == "relu":
        return math.sqrt(2.0)
    if nonlinearity == "leaky_relu":
        if param is None:
            slope = 0.01
        elif not isinstance(param, bool) and isinstance(param, (int, float)):
            slope = param
        else:
            raise ValueError(f"negative_slope {param} not a valid number")
        return math.sqrt(2.0 / (1 + slope ** 2))
    if nonlinearity == "selu":
        return 3.0 / 4
    raise ValueError(f"Unsupported nonlinearity {nonlinearity}")


def kaiming_uniform_(tensor: torch.Tensor, a: float = 0, mode: str = "fan_in", nonlinearity: str = "leaky_relu",
                     reverse: bool = False) -> torch.Tensor:
    fan = _calculate_correct_fan(tensor, mode, reverse)
    k = math.sqrt(3.0) * _calculate_gain(nonlinearity, a) / math.sqrt(fan)
    return uniform_(tensor, -k, k)


def kaiming_normal_(tensor: torch.Tensor, a: float = 0, mode: str = "fan_in", nonlinearity: str = "leaky_relu",
                    reverse: bool = False) -> torch.Tensor:
    fan = _calculate_correct_fan(tensor, mode, reverse)
    return normal_(tensor, 0.0, _calculate_gain(nonlinearity, a) / math.sqrt(fan))


def linear_init_(module) -> None:
    """initializer.py:436-452: a linear layer's default (kaiming-uniform weight with a = sqrt(5); bias U(+-1/sqrt(fan_in)))
    for anything with `.weight` [in, out] and an optional `.bias`."""
    kaiming_uniform_(module.weight, a=math.sqrt(5))
    if getattr(module, "bias", None) is not None:
        fan_in, _ = _calculate_fan_in_and_fan_out(module.weight, reverse=True)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        uniform_(module.bias, -bound, bound)


def conv_init_(module) -> None:
    """initializer.py:455-472: the same for a convolution (`.weight` [cout, cin, ...])."""
    kaiming_uniform_(module.weight, a=math.sqrt(5))
    if getattr(module, "bias", None) is not None:
        fan_in, _ = _calculate_fan_in_and_fan_out(module.weight, reverse=False)
        if fan_in != 0:
            bound = 1 / math.sqrt(fan_in)
            uniform_(module.bias, -bound, bound)


def glorot_normal_(tensor: torch.Tensor) -> torch.Tensor:
    """initializer.py:475-498: jax-style glorot normal -- a standard normal truncated to [-2, 2], scaled by
    sqrt(2 / (fin + fout)) x 0.87962566103423978 (the inverse of the truncated normal's standard deviation)."""
    assert tensor.ndim == 2, f"glorot_normal_ only support 2D tensor now, but got ndim={tensor.ndim}"
    fin, fout = tensor.shape
    stddev = math.sqrt(2.0 / (fin + fout)) * 0.87962566103423978
    trunc_normal_(tensor)
    with torch.no_grad():
        tensor.mul_(stddev)
    return tensor
