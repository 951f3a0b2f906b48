"""TEST INFRASTRUCTURE (oracle) -- never imported by the product path.

Closed-form forward (Taylor-mode) propagation of value / first / second directional
derivative "streams" through a ppsci-style MLP, plus the matching reverse sweep, in numpy
(float64 by default).  This is the *independent* second oracle: it restates the math of
SURVEY.md Appendix A, which is what the HIP kernels implement, and is itself checked
against the reverse-over-reverse restatement of the reference algorithm in
`oracle/ref_torch.py` (tests/test_oracle.py).

Reference behaviour being reproduced (file:line in /root/reference):
  * MLP.forward_tensor           ppsci/arch/mlp.py:281-296   (y = act(y @ W + b), last_fc linear,
                                                              skip_connection quirk: y = 2*y on
                                                              even layers i >= 2 before the act)
  * PeriodEmbedding.forward      ppsci/arch/mlp.py:108-114   (x_k -> [cos(w x_k), sin(w x_k)])
  * Arch.concat_to_tensor        ppsci/arch/base.py:78-112   (inputs concatenated in input_keys order)
  * activations                  ppsci/arch/activation.py:77-88,139-154 (tanh, silu = x*sigmoid(x), sin)
  * jacobian / hessian           ppsci/autodiff/ad.py:56-77,181-236  (per-point d/dx, d2/dx2; here obtained
                                                              as directional derivative streams)

parity unpinned: PaddlePaddle is not installable here, so no output of the real reference
pins the NN numerics; see DESIGN.md section "Oracle".
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class NetSpec:
    """A ppsci.arch.MLP in plain arrays.  weights[l] is [in, out] (Paddle nn.Linear layout)."""

    weights: List[np.ndarray]
    biases: List[np.ndarray]
    activation: str = "tanh"
    # raw-input index -> angular frequency w = 2*pi/period (mlp.py:102)
    periods: Dict[int, float] = field(default_factory=dict)
    skip_connection: bool = False

    @property
    def n_hidden(self) -> int:
        return len(self.weights) - 1

    @property
    def d_raw(self) -> int:
        return self.weights[0].shape[0] - len(self.periods)

    @property
    def d_out(self) -> int:
        return self.weights[-1].shape[1]

    def astype(self, dt) -> "NetSpec":
        return NetSpec(
            [w.astype(dt) for w in self.weights],
            [b.astype(dt) for b in self.biases],
            self.activation,
            dict(self.periods),
            self.skip_connection,
        )

    def zscale(self, layer: int) -> float:
        """mlp.py:286-291: on even hidden layers after the first, `skip = y; y = y + skip`."""
        if self.skip_connection and layer % 2 == 0 and layer >= 2:
            return 2.0
        return 1.0


def make_net(
    d_in: int,
    hidden: Sequence[int],
    d_out: int,
    seed: int = 1234,
    activation: str = "tanh",
    periods: Optional[Dict[int, float]] = None,
    skip_connection: bool = False,
    bias_scale: float = 0.0,
) -> NetSpec:
    """W ~ U(-sqrt(6/(in+out)), +sqrt(6/(in+out))), b = 0 (or small uniform when
    bias_scale > 0 so that bias paths are exercised), `np.random.default_rng(seed)`, drawn layer by
    layer, W then b.  `d_in` is the *raw* input count; period-embedded inputs add one column each."""
    rng = np.random.default_rng(seed)
    periods = dict(periods or {})
    sizes = [d_in + len(periods)] + list(hidden) + [d_out]
    ws, bs = [], []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        lim = np.sqrt(6.0 / (fi + fo))
        ws.append(rng.uniform(-lim, lim, size=(fi, fo)))
        bs.append(rng.uniform(-bias_scale, bias_scale, size=(fo,)) if bias_scale > 0 else np.zeros(fo))
    return NetSpec(ws, bs, activation, periods, skip_connection)


# ----------------------------------------------------------------------------- activations
def act_derivs(name: str, z: np.ndarray):
    """value and first three derivatives of the activation."""
    if name == "tanh":
        s = np.tanh(z)
        d1 = 1.0 - s * s
        d2 = -2.0 * s * d1
        d3 = d1 * (6.0 * s * s - 2.0)
        return s, d1, d2, d3
    if name in ("silu", "swish_fixed"):
        g = 1.0 / (1.0 + np.exp(-z))
        g1 = g * (1.0 - g)
        s = z * g
        d1 = g + z * g1
        d2 = g1 * (2.0 + z * (1.0 - 2.0 * g))
        d3 = 3.0 * g1 * (1.0 - 2.0 * g) + z * (g1 * (1.0 - 2.0 * g) ** 2 - 2.0 * g1 * g1)
        return s, d1, d2, d3
    if name == "sin":
        s, c = np.sin(z), np.cos(z)
        return s, c, -s, -c
    raise ValueError(f"unsupported activation {name}")


# ----------------------------------------------------------------------------- input streams
def input_streams(net: NetSpec, X: np.ndarray, dirs: np.ndarray, n2: int):
    """Embedded input h0 and its directional streams.

    X [N, d_raw]; dirs [n1, d_raw]; second-order streams are taken along dirs[:n2].
    Returns list of S = 1 + n1 + n2 arrays [N, d0] in the order (value, first..., second...).
    """
    N = X.shape[0]
    n1 = dirs.shape[0]
    cols_v, cols_1, cols_2 = [], [[] for _ in range(n1)], [[] for _ in range(n2)]
    for j in range(net.d_raw):
        x = X[:, j]
        if j in net.periods:
            w = net.periods[j]
            c, s = np.cos(w * x), np.sin(w * x)
            cols_v += [c, s]
            for i in range(n1):
                v = dirs[i, j]
                cols_1[i] += [-w * s * v, w * c * v]
            for i in range(n2):
                v = dirs[i, j]
                cols_2[i] += [-w * w * c * v * v, -w * w * s * v * v]
        else:
            cols_v.append(x)
            for i in range(n1):
                cols_1[i].append(np.full(N, dirs[i, j], dtype=X.dtype))
            for i in range(n2):
                cols_2[i].append(np.zeros(N, dtype=X.dtype))
    st = [np.stack(cols_v, 1)]
    st += [np.stack(c, 1) for c in cols_1]
    st += [np.stack(c, 1) for c in cols_2]
    return st


# ----------------------------------------------------------------------------- forward
def taylor_forward(net: NetSpec, X: np.ndarray, dirs: np.ndarray, n2: int, keep: bool = False):
    """Returns U [m, S, N] (S = 1+n1+n2) and, if keep, the per-layer caches for the reverse sweep."""
    dirs = np.asarray(dirs, dtype=X.dtype).reshape(-1, net.d_raw)
    n1 = dirs.shape[0]
    assert 0 <= n2 <= n1
    h = input_streams(net, X, dirs, n2)
    cache = {"h": [h], "z": [], "n1": n1, "n2": n2}
    for l in range(net.n_hidden):
        W, b, c = net.weights[l], net.biases[l], net.zscale(l)
        z = [c * (h[0] @ W + b)] + [c * (hs @ W) for hs in h[1:]]
        s, d1, d2, _ = act_derivs(net.activation, z[0])
        hn = [s]
        for i in range(n1):
            hn.append(d1 * z[1 + i])
        for i in range(n2):
            hn.append(d2 * z[1 + i] * z[1 + i] + d1 * z[1 + n1 + i])
        cache["z"].append(z)
        cache["h"].append(hn)
        h = hn
    W, b = net.weights[-1], net.biases[-1]
    out = [h[0] @ W + b] + [hs @ W for hs in h[1:]]
    U = np.stack([o.T for o in out], axis=1)  # [m, S, N]
    return (U, cache) if keep else U


# ----------------------------------------------------------------------------- backward
def taylor_backward(net: NetSpec, cache, Ubar: np.ndarray):
    """Reverse sweep.  Ubar [m, S, N] = dL/dU.  Returns (gW list, gb list) shaped like net."""
    n1, n2 = cache["n1"], cache["n2"]
    S = 1 + n1 + n2
    L = net.n_hidden
    gW = [np.zeros_like(w) for w in net.weights]
    gb = [np.zeros_like(b) for b in net.biases]
    hL = cache["h"][L]
    ub = [Ubar[:, s, :].T for s in range(S)]  # [N, m] each
    for s in range(S):
        gW[L] += hL[s].T @ ub[s]
    gb[L] += ub[0].sum(0)
    hbar = [u @ net.weights[L].T for u in ub]
    for l in range(L - 1, -1, -1):
        z = cache["z"][l]
        c = net.zscale(l)
        _, d1, d2, d3 = act_derivs(net.activation, z[0])
        zb = [None] * S
        acc = d1 * hbar[0]
        for i in range(n1):
            zi = z[1 + i]
            zbi = d1 * hbar[1 + i]
            acc = acc + d2 * zi * hbar[1 + i]
            if i < n2:
                zii = z[1 + n1 + i]
                hb2 = hbar[1 + n1 + i]
                zb[1 + n1 + i] = d1 * hb2
                zbi = zbi + 2.0 * d2 * zi * hb2
                acc = acc + (d3 * zi * zi + d2 * zii) * hb2
            zb[1 + i] = zbi
        zb[0] = acc
        zb = [c * t for t in zb]
        hp = cache["h"][l]
        for s in range(S):
            gW[l] += hp[s].T @ zb[s]
        gb[l] += zb[0].sum(0)
        if l > 0:
            hbar = [t @ net.weights[l].T for t in zb]
    return gW, gb


def flat_params(net: NetSpec) -> np.ndarray:
    """Flat parameter vector in model.parameters() order: W0, b0, W1, b1, ... (mlp.py:264-277)."""
    return np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(net.weights, net.biases)])


def flat_grads(gW, gb) -> np.ndarray:
    return np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(gW, gb)])
