"""Activation registry (/root/reference/ppsci/arch/activation.py:139-154).  Only the activations
with a fused HIP implementation are accepted on the hot path: tanh, silu (= x*sigmoid(x), :77-88),
sin, cos, sigmoid, gelu (exact, erf), siren (= sin(30 x), :91-136, with its own weight initialisation), swish
(x*sigmoid(beta x), trainable scalar beta per layer, :49-58), stan (tanh(x)(1 + beta x), trainable beta[H], :28-46) and
the piecewise-linear / exponential-linear family relu, leaky_relu (slope 0.01), elu (alpha 1), selu, identity -- i.e. every
entry of the reference's act_func_dict."""
HIP_ACTIVATIONS = ("tanh", "silu", "sin", "sigmoid", "cos", "gelu", "siren", "swish", "stan", "relu", "leaky_relu", "elu",
                   "selu", "identity")
REFERENCE_ACTIVATIONS = ("elu", "relu", "selu", "gelu", "leaky_relu", "sigmoid", "silu", "sin", "cos", "swish",
                         "tanh", "identity", "siren", "stan")


def get_activation(name: str) -> str:
    low = name.lower()
    if low not in REFERENCE_ACTIVATIONS:
        raise ValueError(f"Act name must be in {REFERENCE_ACTIVATIONS}, but got {name}")
    if low not in HIP_ACTIVATIONS:
        raise NotImplementedError(f"activation {name!r} has no fused HIP kernel yet (available: {HIP_ACTIVATIONS})")
    return low
