"""Symbolic per-point expression graph: the front-end that lowers a constraint's expressions to
(derivative stream set, epilogue program) for the fused HIP kernels.

Why it exists: the reference evaluates `model(x)`, `jacobian(u, x)`, `hessian(u, x)` and the
operator tree eagerly on paddle tensors with a dynamic autograd graph
(/root/reference/ppsci/autodiff/ad.py, ppsci/utils/symbolic.py:184-504, expression.py:96-102).
Here the same Python calls are made ONCE on proxy tensors (`Sym`); the recorded graph is then
compiled into
  * the set of network derivatives the Taylor-mode kernel must carry (hotpath.StreamSpec), and
  * a postfix program for the epilogue VM (hotpath.Program), incl. the MSE terms of
    ppsci/loss/mse.py:82-105.
Derivatives of arbitrary recorded expressions are obtained by symbolic differentiation down to the
network-output leaves (`net` -> `der`), which is what repeated `paddle.grad(..., create_graph=True)`
computes numerically (ad.py:73-75).
"""
from __future__ import annotations

import contextlib
import zlib
import math
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib as L
from . import hotpath as hp

Number = Union[int, float]

_UNARY_OPS = {
    "sin": L.OP_SIN, "cos": L.OP_COS, "tanh": L.OP_TANH, "exp": L.OP_EXP, "log": L.OP_LOG, "sqrt": L.OP_SQRT,
    "abs": L.OP_ABS, "sinh": L.OP_SINH, "cosh": L.OP_COSH, "tan": L.OP_TAN, "neg": L.OP_NEG, "sign": L.OP_SIGN,
    "heaviside": L.OP_HEAVISIDE, "detach": L.OP_DETACH,
    "asin": L.OP_ASIN, "acos": L.OP_ACOS, "atan": L.OP_ATAN, "asinh": L.OP_ASINH, "acosh": L.OP_ACOSH,
    "atanh": L.OP_ATANH, "erf": L.OP_ERF, "lgamma": L.OP_LGAMMA, "ceil": L.OP_CEIL, "floor": L.OP_FLOOR,
}
_BINARY_OPS = {"add": L.OP_ADD, "sub": L.OP_SUB, "mul": L.OP_MUL, "div": L.OP_DIV, "pow": L.OP_POW,
               "max": L.OP_MAX, "min": L.OP_MIN, "atan2": L.OP_ATAN2}


class Sym:
    """A per-point scalar field ([N, 1] in the reference).  Immutable; hash-consed by structure."""

    __slots__ = ("kind", "name", "value", "model", "comp", "dirs", "op", "args", "_key")
    _cache: Dict[tuple, "Sym"] = {}

    def __new__(cls, kind, name=None, value=None, model=None, comp=None, dirs=(), op=None, args=()):
        key = (kind, name, None if value is None else float(np.float32(value)), id(model) if model is not None else None,
               comp, tuple(dirs), op, tuple(id(a) for a in args))
        hit = cls._cache.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.kind, self.name, self.model, self.comp = kind, name, model, comp
        self.value = None if value is None else float(np.float32(value))  # ConstantNode: fp32 (symbolic.py:448-454)
        self.dirs, self.op, self.args = tuple(dirs), op, tuple(args)
        self._key = key
        cls._cache[key] = self
        return self

    # ---- constructors
    @staticmethod
    def input(name: str) -> "Sym":
        return Sym("in", name=name)

    @staticmethod
    def aux(name: str) -> "Sym":
        return Sym("aux", name=name)

    @staticmethod
    def param(name: str, slot: int) -> "Sym":
        """A learnable equation parameter (ParameterNode, symbolic.py:471-485): one scalar for all points."""
        return Sym("param", name=name, comp=int(slot))

    @staticmethod
    def const(v: Number) -> "Sym":
        return Sym("const", value=float(v))

    @staticmethod
    def net(model, comp: int, dirs: Sequence[str] = ()) -> "Sym":
        return Sym("net", model=model, comp=comp, dirs=tuple(sorted(dirs)))

    # ---- reference-tensor look-alike surface
    @property
    def shape(self):
        return [-1, 1]

    def detach(self) -> "Sym":
        return apply("detach", self)

    def __repr__(self):
        if self.kind == "in":
            return self.name
        if self.kind == "aux":
            return f"aux:{self.name}"
        if self.kind == "param":
            return f"param:{self.name}"
        if self.kind == "const":
            return repr(self.value)
        if self.kind == "net":
            k = self.model.output_keys[self.comp]
            return k + "".join(f"__{d}" for d in self.dirs)
        if self.kind == "rows":
            return f"{self.args[0]!r}[{self.comp[0]}:{'' if self.comp[1] is None else self.comp[1]}]"
        return f"{self.op}({', '.join(map(repr, self.args))})"

    # ---- arithmetic
    def __add__(self, o): return apply("add", self, _lift(o))
    def __radd__(self, o): return apply("add", _lift(o), self)
    def __sub__(self, o): return apply("sub", self, _lift(o))
    def __rsub__(self, o): return apply("sub", _lift(o), self)
    def __mul__(self, o): return apply("mul", self, _lift(o))
    def __rmul__(self, o): return apply("mul", _lift(o), self)
    def __truediv__(self, o): return apply("div", self, _lift(o))
    def __rtruediv__(self, o): return apply("div", _lift(o), self)
    def __pow__(self, o): return apply("pow", self, _lift(o))
    def __rpow__(self, o): return apply("pow", _lift(o), self)
    def __neg__(self): return apply("neg", self)
    def __pos__(self): return self

    # ---- values at trace time: data-dependent Python control flow (see `batch_values`)
    def _scalar(self, what: str):
        v = concrete_values(self, what)
        if v.size != 1:
            raise TypeError(f"{what} of a traced expression with {v.size} values: only one-element values convert to Python scalars")
        _TRACE.concretized.append(f"{what}({self!r}) = {float(v.reshape(-1)[0])!r}")
        return v.reshape(-1)[0]

    def __bool__(self):
        return bool(self._scalar("bool") != 0)

    def __float__(self):
        return float(self._scalar("float"))

    def __int__(self):
        return int(self._scalar("int"))

    def item(self):
        return float(self._scalar("item"))

    def _indicator(self, o, what):
        """The comparison as a traced per-point 0 / 1 value (what the reference's bool tensor stands for in
        `paddle.where(cond, a, b)`; examples/chip_heat/chip_heat.py:217-232): built from the VM's heaviside (x > 0 ? 1 : 0)
        and abs, derivative zero."""
        d = self - _lift(o)
        one = Sym.const(1.0)
        if what == "gt":
            return apply("heaviside", d)
        if what == "lt":
            return apply("heaviside", -d)
        if what == "ge":
            return one - apply("heaviside", -d)
        if what == "le":
            return one - apply("heaviside", d)
        if what == "eq":
            return one - apply("heaviside", apply("abs", d))
        return apply("heaviside", apply("abs", d))  # ne

    def _compare(self, o, f, what):
        if not isinstance(o, (Sym, int, float, np.floating, np.integer)):
            return NotImplemented
        if _TRACE.values is None:
            # No fixed batch to look at: the comparison itself is traced.  bool() of it still raises (see _scalar).
            return self._indicator(o, what)
        try:
            ov = concrete_values(o, what) if isinstance(o, Sym) else np.float32(o)
            v = f(concrete_values(self, what), ov)
        except TypeError:
            # a fixed batch, but the compared values depend on the network: no answer at trace time, a traced indicator
            return self._indicator(o, what)
        # (the ANSWER is part of the record: ranks of a data-parallel job see different shards and must not silently compile
        # different programs, solver.Solver._check_trace_decisions)
        if v.size == 1:
            ans = bool(v.reshape(-1)[0])
            _TRACE.concretized.append(f"{what}({self!r}) = {ans!r}")
            return ans
        # One answer per point of the fixed batch.  As a Python value it is the bool array the reference's tensor holds (its
        # use -- indexing, any(), all() -- specialises the trace to this batch, so it is recorded when it is LOOKED at);
        # handed to functional.where it is the traced indicator, which holds for any batch and specialises nothing.
        return _BatchMask(v, self, o, what)

    def __lt__(self, o): return self._compare(o, np.less, "lt")
    def __le__(self, o): return self._compare(o, np.less_equal, "le")
    def __gt__(self, o): return self._compare(o, np.greater, "gt")
    def __ge__(self, o): return self._compare(o, np.greater_equal, "ge")

    def __eq__(self, o):
        # two traced values: as a Python truth value, identity (nodes are hash-consed: the same structure IS the same object --
        # what `in` / dict lookups of this module rely on); as the condition of functional.where, the per-point comparison.
        # A number: the comparison of the reference's tensors, on the batch the trace is specialised to
        if isinstance(o, Sym):
            return _IdentityAnswer(self is o, self, o, "eq")
        return self._compare(o, np.equal, "eq")

    def __ne__(self, o):
        if isinstance(o, Sym):
            return _IdentityAnswer(self is not o, self, o, "ne")
        return self._compare(o, np.not_equal, "ne")

    __hash__ = object.__hash__

    def __getitem__(self, key):
        """`expr[a:b]`: ROWS a..b of the batch (examples/euler_beam/euler_beam.py:49-54 picks the boundary point each condition
        belongs to).  Legal as the whole of an output expression -- compile.CompiledConstraint turns a one-row slice into a
        per-point weight mask (see there) -- or under float() / bool() / a comparison (its value at trace time).  Anything
        else with it raises."""
        if isinstance(key, (int, np.integer)):  # x[k]: the reference's [1]-shaped row k
            key = slice(int(key), int(key) + 1 if int(key) != -1 else None)
        if isinstance(key, slice) and key.step in (None, 1) and all(isinstance(v, (int, type(None))) for v in (key.start, key.stop)):
            return Sym("rows", comp=(key.start or 0, key.stop), args=(self,))
        raise TypeError("a traced expression can only be indexed by a row x[k] or a contiguous row slice x[a:b]")

    # ---- batch reductions (tensor.mean() / .sum() in a user expression: utils/expression.py:96-102 runs arbitrary tensor code)
    def _reduce(self, op: str, axis, keepdim) -> "Sym":
        if axis not in (None, 0, (0,), [0], (0, 1), [0, 1]):
            raise NotImplementedError(f"{op}(axis={axis!r}) of a traced [N, 1] value: only the reduction over the batch")
        if self.kind == "rows":
            raise TypeError("a batch reduction of a row slice is not lowered (reduce the whole column)")
        if self.kind == "reduce" or (op == "mean" and self.kind == "const"):
            return self  # (already one scalar for the whole batch)
        return Sym("reduce", op=op, args=(self,))

    def mean(self, axis=None, keepdim=False):
        """One scalar for the whole batch (broadcast back over the points wherever it is used).  Lowered as a two-pass
        program (graph.lower): the sum is taken by a first launch and read as a parameter slot by the residual program;
        its adjoint is carried back to every point by a third launch.  Over the GLOBAL batch under data parallelism."""
        return self._reduce("mean", axis, keepdim)

    def sum(self, axis=None, keepdim=False):  # noqa: A003
        return self._reduce("sum", axis, keepdim)

    def sin(self): return apply("sin", self)
    def cos(self): return apply("cos", self)
    def tanh(self): return apply("tanh", self)
    def exp(self): return apply("exp", self)
    def log(self): return apply("log", self)
    def sqrt(self): return apply("sqrt", self)
    def abs(self): return apply("abs", self)
    def pow(self, o): return apply("pow", self, _lift(o))


# ---- the batch a trace is specialised to ------------------------------------------------------------------------------
# The reference evaluates expressions on real tensors, so Python may branch on their values (`if float(d["x"][0]) == 0:`,
# utils/expression.py:96-102 just calls the user function).  A trace cannot follow a value that changes from step to step;
# but a constraint whose batch is FIXED (full batch, no shuffling: the boundary / initial-condition sets such code looks
# at) has one answer for every step.  `batch_values(...)` gives the trace those values: float() / bool() / comparisons
# of expressions of the INPUT columns evaluate on them (fp32, the reference's dtype), the branch taken is the one the
# reference takes on every step, and the trace stays a per-point program for the kernels.  Values that depend on the
# network (they change as it trains) still raise.  `_TRACE.concretized` records what was asked, so that the caller can
# refuse to reuse such a trace for a batch with other values.
class _IdentityAnswer:
    """`symA == symB` / `symA != symB`: truthy by node identity; `.indicator()` is the traced per-point comparison."""
    __slots__ = ("_ans", "_a", "_b", "_what")

    def __init__(self, ans: bool, a: "Sym", b: "Sym", what: str):
        self._ans, self._a, self._b, self._what = bool(ans), a, b, what

    def __bool__(self):
        return self._ans

    def __eq__(self, other):
        return self._ans == bool(other)

    def __hash__(self):
        return hash(self._ans)

    def __repr__(self):
        return repr(self._ans)

    def indicator(self) -> "Sym":
        return self._a._indicator(self._b, self._what)


class _BatchMask(np.ndarray):
    """The per-point answers of a comparison on the fixed batch of a static constraint (a bool array), which also
    remembers the comparison: `functional.where` takes the traced indicator instead of the values."""

    def __new__(cls, values: np.ndarray, a: "Sym", b, what: str):
        obj = np.asarray(values, dtype=bool).view(cls)
        obj._cmp = (a, b, what)
        obj._noted = False
        return obj

    def __array_finalize__(self, obj):
        self._cmp = getattr(obj, "_cmp", None)
        self._noted = getattr(obj, "_noted", True)

    def _note(self):
        if not self._noted and self._cmp is not None:
            a, _, what = self._cmp
            v = np.asarray(self)
            _TRACE.concretized.append(f"{what}({a!r}) = {int(v.sum())} of {v.size} true, crc {zlib.crc32(np.packbits(v).tobytes()):08x}")
            self._noted = True

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        for x in inputs:
            if isinstance(x, _BatchMask):
                x._note()
        inputs = tuple(np.asarray(x) if isinstance(x, _BatchMask) else x for x in inputs)
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        self._note()
        strip = lambda x: np.asarray(x) if isinstance(x, _BatchMask) else x  # noqa: E731
        return func(*[strip(a) for a in args], **{k: strip(v) for k, v in kwargs.items()})

    def __getitem__(self, key):
        self._note()
        return np.asarray(self)[key]

    def __bool__(self):
        self._note()
        return bool(np.asarray(self))

    def __iter__(self):
        self._note()
        return iter(np.asarray(self))

    def tolist(self):
        self._note()
        return np.asarray(self).tolist()

    def indicator(self) -> "Sym":
        a, b, what = self._cmp
        return a._indicator(b, what)


class _TraceState:
    values: Optional[Dict[str, np.ndarray]] = None
    concretized: List[str] = []


_TRACE = _TraceState()


@contextlib.contextmanager
def batch_values(values: Optional[Dict[str, object]]):
    """values: column name -> array of the bound batch (None: no specialisation, value requests raise)."""
    def host(a):
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        return a.astype(np.float32).reshape(-1)

    prev = (_TRACE.values, _TRACE.concretized)
    _TRACE.values = None if values is None else {k: host(v) for k, v in values.items()}
    _TRACE.concretized = []
    try:
        yield _TRACE
    finally:
        _TRACE.values, _TRACE.concretized = prev


_NUMPY_OPS = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "pow": np.power, "max": np.maximum,
              "min": np.minimum, "atan2": np.arctan2, "sin": np.sin, "cos": np.cos, "tanh": np.tanh, "exp": np.exp,
              "log": np.log, "sqrt": np.sqrt, "abs": np.abs, "sinh": np.sinh, "cosh": np.cosh, "tan": np.tan,
              "neg": np.negative, "sign": np.sign, "detach": lambda v: v, "asin": np.arcsin, "acos": np.arccos,
              "atan": np.arctan, "asinh": np.arcsinh, "acosh": np.arccosh, "atanh": np.arctanh, "ceil": np.ceil,
              "floor": np.floor}


def concrete_values(s: "Sym", what: str = "value") -> np.ndarray:
    """fp32 values of an expression of the input columns on the batch the trace is specialised to."""
    if _TRACE.values is None:
        raise TypeError(f"{what}() of the traced expression {s!r}: data-dependent Python control flow can only be followed for a "
                        "constraint whose batch is fixed (one full batch, no shuffling); this one changes every iteration")
    if s.kind in ("in", "aux"):
        if s.name not in _TRACE.values:
            raise TypeError(f"{what}() of {s!r}: the bound batch has no column {s.name!r}")
        return _TRACE.values[s.name]
    if s.kind == "const":
        return np.float32(s.value).reshape(1)
    if s.kind == "rows":
        a, b = s.comp
        return concrete_values(s.args[0], what)[a:b]
    if s.kind == "reduce":
        v = concrete_values(s.args[0], what).astype(np.float32)
        return np.asarray(v.mean(dtype=np.float32) if s.op == "mean" else v.sum(dtype=np.float32), dtype=np.float32).reshape(1)
    if s.kind == "op" and s.op in _NUMPY_OPS:
        with np.errstate(all="ignore"):
            return np.asarray(_NUMPY_OPS[s.op](*[concrete_values(a, what) for a in s.args]), dtype=np.float32)
    raise TypeError(f"{what}() of the traced expression {s!r}: it depends on the network (or a learnable parameter), whose value "
                    "changes every step -- data-dependent Python control flow on it cannot be lowered to the fused HIP path")


# ---- batch coupling by a constant matrix ------------------------------------------------------------------------------
# `lhs[:len(rhs)] - paddle.mm(int_mat, v)` (ppsci/equation/ide/volterra.py:66-77): rows i < R of the batch get the residual
# lhs_i - sum_q M[i, q] v_q, the other rows (the quadrature points) carry no residual.  Only legal as a WHOLE output expression:
# lower() turns it into a per-point program around two small matrix-vector launches (engine.FusedConstraint._forward_couplings).
COUPLE_RHS_PREFIX, COUPLE_VBAR_PREFIX = "__couple_rhs__", "__couple_vbar__"
_COUPLE_MATS: Dict[str, np.ndarray] = {}


def couple(lhs, matrix, v) -> "Sym":
    M = np.ascontiguousarray(np.asarray(matrix, dtype=np.float32))
    if M.ndim != 2:
        raise ValueError(f"couple(): a [rows, batch] matrix is needed, got shape {M.shape}")
    name = f"couple{len(_COUPLE_MATS)}_{zlib.crc32(M.tobytes()):08x}"
    _COUPLE_MATS[name] = M
    return Sym("couple", name=name, comp=(int(M.shape[0]), int(M.shape[1])), args=(_lift(lhs), _lift(v)))


def _lift(v) -> Sym:
    if isinstance(v, Sym):
        return v
    if isinstance(v, (int, float, np.floating, np.integer)):
        return Sym.const(float(v))
    if isinstance(v, np.ndarray) and v.size == 1:
        return Sym.const(float(v.reshape(-1)[0]))
    raise TypeError(f"cannot mix a traced expression with {type(v)}")


def is_const(s: Sym, v: Optional[float] = None) -> bool:
    return s.kind == "const" and (v is None or s.value == v)


def apply(op: str, *args: Sym) -> Sym:
    """Builds an op node.  Only exact algebraic identities with 0/1 are folded (they do not change the
    fp32 result), so the reference's operation order is preserved."""
    if any(a.kind == "rows" for a in args):
        raise TypeError("arithmetic on a row slice of the batch is not a per-point program")
    if any(a.kind == "couple" for a in args):
        raise TypeError("a batch-coupled residual (graph.couple) must be the whole output expression")
    if op in _BINARY_OPS:
        a, b = args
        if op == "add":
            if is_const(a, 0.0):
                return b
            if is_const(b, 0.0):
                return a
        elif op == "sub":
            if is_const(b, 0.0):
                return a
        elif op == "mul":
            if is_const(a, 0.0) or is_const(b, 0.0):
                return Sym.const(0.0)
            if is_const(a, 1.0):
                return b
            if is_const(b, 1.0):
                return a
        elif op == "div":
            if is_const(b, 1.0):
                return a
        elif op == "pow":
            if is_const(b, 1.0):
                return a
        if a.kind == "const" and b.kind == "const" and op in ("add", "sub", "mul"):
            f = {"add": np.add, "sub": np.subtract, "mul": np.multiply}[op]
            return Sym.const(float(f(np.float32(a.value), np.float32(b.value))))
        return Sym("op", op=op, args=(a, b))
    if op in _UNARY_OPS:
        (a,) = args
        if op == "neg" and a.kind == "const":
            return Sym.const(-a.value)
        return Sym("op", op=op, args=(a,))
    raise NotImplementedError(f"operation {op!r} is not supported by the epilogue VM")


# ----------------------------------------------------------------------------- differentiation
def net_raw_vars(model) -> Tuple[str, ...]:
    """The data-dict variables a network's outputs depend on: its input keys, or -- with a registered input
    transform -- the raw variables its input features are built from."""
    feats = getattr(model, "_traced_features", None)
    if not feats:
        return tuple(model.input_keys)
    names: List[str] = []
    for n in _walk(list(feats.values())):
        if n.kind in ("in", "aux") and n.name not in names:
            names.append(n.name)
    return tuple(names)


def directional(e: Sym, d, order: int) -> Sym:
    """order-th derivative of e along direction d: a variable name, a pair (a, b) for the direction a + b, or a triple
    (a, b, -1.0) for a - b."""
    if isinstance(d, tuple):
        a, b = d[0], d[1]
        sgn = float(d[2]) if len(d) > 2 else 1.0
        if order <= 2 and sgn == 1.0:
            if order == 1:
                return diff(e, a) + diff(e, b)
            ea = diff(e, a)
            return diff(ea, a) + 2.0 * diff(ea, b) + diff(diff(e, b), b)
        out = e
        for _ in range(order):
            out = diff(out, a) + sgn * diff(out, b)
        return out
    out = e
    for _ in range(order):
        out = diff(out, d)
    return out


def diff(e: Sym, var: str) -> Sym:
    """d e / d var, symbolically (what jacobian() obtains numerically in the reference)."""
    k = e.kind
    if k == "in":
        return Sym.const(1.0 if e.name == var else 0.0)
    if k == "aux":  # a dataset column that is not a network input (raw variable of an input transform, nu, sdf, ...)
        return Sym.const(1.0 if e.name == var else 0.0)
    if k in ("const", "param"):
        return Sym.const(0.0)
    if k == "couple":
        raise NotImplementedError("derivative of a batch-coupled residual w.r.t. a per-point variable")
    if k == "reduce":
        raise NotImplementedError(f"derivative of the batch reduction {e!r} w.r.t. the per-point variable {var!r}: differentiate "
                                  "first, reduce afterwards")
    if k == "net":
        if var not in net_raw_vars(e.model):
            return Sym.const(0.0)
        if len(e.dirs) >= 4:
            raise NotImplementedError(
                f"derivative order {len(e.dirs) + 1} of a network output ({e!r} w.r.t. {var}) is beyond the fused "
                "HIP kernels (orders 0..4)")
        return Sym.net(e.model, e.comp, e.dirs + (var,))
    op, a = e.op, e.args
    if op == "detach":
        return Sym.const(0.0)
    if op == "add":
        return diff(a[0], var) + diff(a[1], var)
    if op == "sub":
        return diff(a[0], var) - diff(a[1], var)
    if op == "mul":
        return diff(a[0], var) * a[1] + a[0] * diff(a[1], var)
    if op == "div":
        return diff(a[0], var) / a[1] - e * diff(a[1], var) / a[1]
    if op == "neg":
        return -diff(a[0], var)
    if op == "pow":
        da, db = diff(a[0], var), diff(a[1], var)
        out = Sym.const(0.0)
        if not is_const(da, 0.0):
            out = out + a[1] * apply("pow", a[0], a[1] - 1.0) * da
        if not is_const(db, 0.0):
            out = out + e * apply("log", a[0]) * db
        return out
    if op == "atan2":  # atan2(y, x)
        den = a[0] * a[0] + a[1] * a[1]
        return (a[1] * diff(a[0], var) - a[0] * diff(a[1], var)) / den
    d = diff(a[0], var)
    if is_const(d, 0.0):
        return d
    if op == "asin":
        return d / apply("sqrt", 1.0 - a[0] * a[0])
    if op == "acos":
        return -(d / apply("sqrt", 1.0 - a[0] * a[0]))
    if op == "atan":
        return d / (1.0 + a[0] * a[0])
    if op == "asinh":
        return d / apply("sqrt", a[0] * a[0] + 1.0)
    if op == "acosh":
        return d / apply("sqrt", a[0] * a[0] - 1.0)
    if op == "atanh":
        return d / (1.0 - a[0] * a[0])
    if op == "erf":
        return apply("exp", -(a[0] * a[0])) * (2.0 / math.sqrt(math.pi)) * d
    if op == "sin":
        return apply("cos", a[0]) * d
    if op == "cos":
        return -(apply("sin", a[0]) * d)
    if op == "tanh":
        return (1.0 - e * e) * d
    if op == "exp":
        return e * d
    if op == "log":
        return d / a[0]
    if op == "sqrt":
        return d / (2.0 * e)
    if op == "abs":
        return apply("sign", a[0]) * d
    if op == "sinh":
        return apply("cosh", a[0]) * d
    if op == "cosh":
        return apply("sinh", a[0]) * d
    if op == "tan":
        return (1.0 + e * e) * d
    if op in ("sign", "heaviside", "ceil", "floor"):
        return Sym.const(0.0)
    raise NotImplementedError(f"derivative of {op!r}")


# ----------------------------------------------------------------------------- lowering
def _walk(roots: Iterable[Sym]) -> List[Sym]:
    seen, order = set(), []

    def rec(n: Sym):
        if id(n) in seen:
            return
        seen.add(id(n))
        for a in n.args:
            rec(a)
        order.append(n)

    for r in roots:
        rec(r)
    return order


# (n1, n2, n3, n4) combinations the kernels are instantiated for (taylor_fwd.inc / taylor_bwd.inc): first-order
# directions, and how many of the FIRST of them also carry second / third / fourth-order streams
_INSTANTIATED = [(0, 0, 0, 0), (1, 1, 0, 0), (2, 0, 0, 0), (2, 1, 0, 0), (2, 2, 0, 0), (3, 2, 0, 0), (3, 3, 0, 0),
                 (4, 3, 0, 0),  # unsteady 3-D NavierStokes: first derivatives along x, y, z, t, second along x, y, z
                 (1, 1, 1, 0), (1, 1, 1, 1), (2, 2, 2, 2), (4, 4, 4, 4)]


class Lowered:
    """Result of lowering: everything engine.FusedConstraint needs except the device arrays."""

    def __init__(self, model, streams: hp.StreamSpec, program: hp.Program, input_names: List[str],
                 aux_names: List[str], loss_keys: List[str], value_index: Dict[str, int]):
        self.model, self.streams, self.program = model, streams, program
        self.input_names, self.aux_names, self.loss_keys = input_names, aux_names, loss_keys
        self.value_index = value_index  # output name -> program value index
        self.causal: List[tuple] = []
        self.periodic: List[tuple] = []
        self.param_slots: List[int] = []  # slots of the learnable equation parameters the program reads
        self.couplings: Optional[dict] = None  # batch couplings: {"items": [...], "pv": value program, "p3": adjoint program}
        self.reductions: Optional[dict] = None  # batch reductions: {"k": count, "p1": summand program, "p3": adjoint program}
        self.nets: List[tuple] = []  # (model, StreamSpec, first U row, input indices) per network of the constraint


def lower(outputs: Dict[str, Sym], losses: Sequence[dict], extra_outputs: Sequence[str] = (),
          n_global: Optional[int] = None) -> Lowered:
    """outputs: name -> traced expression.  losses: dicts with keys
         key (output name), label (aux name or None), weight (aux name or None), area (aux name or None), scale.
       Residual row k of the epilogue corresponds to losses[k]; `extra_outputs` are appended as
       scale-0 residual rows so that eval / predict can read their values."""
    roots = list(outputs.values())
    nodes = _walk(roots)
    # batch reductions (Sym.mean / Sym.sum): reduction k is read from parameter slot k by the residual program
    reduces = [n for n in nodes if n.kind == "reduce"]
    if reduces:
        if any(n.kind == "param" for n in nodes):
            raise NotImplementedError("batch reductions together with learnable equation parameters (both use the parameter slots)")
        if len(reduces) > L.MAX_EPARAM:
            raise NotImplementedError(f"more than {L.MAX_EPARAM} batch reductions in the expressions of one constraint")
        if any(m.kind == "reduce" for r in reduces for m in _walk([r.args[0]])):
            raise NotImplementedError("a batch reduction inside a batch reduction (e.g. a variance written with two means): "
                                      "one level is lowered")
        if any(r.op == "mean" for r in reduces) and not n_global:
            raise NotImplementedError("mean() over the batch needs the global batch size")
    red_slot = {id(r): k for k, r in enumerate(reduces)}
    couples = [n for n in nodes if n.kind == "couple"]
    if couples:
        if reduces:
            raise NotImplementedError("batch couplings together with batch reductions")
        if any(id(c) not in {id(r) for r in roots} for c in couples) or any(a.kind == "couple" for n in nodes for a in n.args):
            raise NotImplementedError("a batch-coupled residual must be the whole output expression")
    models = {id(n.model): n.model for n in nodes if n.kind == "net"}  # ModelList members: in order of appearance
    model_list = list(models.values())
    model = model_list[0] if model_list else None

    # ---- derivative set -> stream specification
    # order[d] = highest pure derivative order needed along direction d (a variable, (a, b) = a + b, (a, b, -1) = a - b)
    order: Dict[object, int] = {}
    mixed, mixed4 = set(), set()

    def need(d, k):
        order[d] = max(order.get(d, 0), k)

    for n in nodes:
        if n.kind != "net" or not n.dirs:
            continue
        vs = sorted(set(n.dirs))
        k = len(n.dirs)
        if len(vs) == 1:
            need(vs[0], k)
        elif k == 2:
            a, b = sorted(n.dirs)  # (a + b and b + a are ONE direction: one key)
            mixed.add((a, b))
            need(a, 2), need(b, 2), need((a, b), 2)
        elif k == 4 and len(vs) == 2 and n.dirs.count(vs[0]) == 2:
            a, b = vs  # u_aabb = (D4_{a+b} + D4_{a-b} - 2 D4_a - 2 D4_b) / 12
            mixed4.add((a, b))
            need(a, 4), need(b, 4), need((a, b), 4), need((a, b, -1.0), 4)
        elif k == 3 and len(vs) == 2:
            # u_aab (a twice, b once) = (D3_{a+b} - D3_{a-b} - 2 D3_b) / 6: D3_{a+-b} = u_aaa +- 3 u_aab + 3 u_abb +- u_bbb
            # (the shear forces of examples/biharmonic2d/biharmonic2d.py:325-336: d/dx (u_xx + u_yy))
            a, b = (vs[0], vs[1]) if n.dirs.count(vs[0]) == 2 else (vs[1], vs[0])
            lo, hi = sorted((a, b))  # (D3 along b - a is minus D3 along a - b: one stream serves u_aab and u_abb)
            need(b, 3), need((lo, hi), 3), need((lo, hi, -1.0), 3)
        else:
            raise NotImplementedError(
                f"mixed derivative {n!r}: beyond pure derivatives the fused HIP kernels carry u_ab, u_aab and u_aabb")
    in_keys: List[str] = []  # union of the members' inputs: the constraint's input arrays
    for mm in model_list:
        in_keys += [k for k in net_raw_vars(mm) if k not in in_keys]

    def rank(d):  # variables in input order first, then the combined directions
        return (0, in_keys.index(d)) if isinstance(d, str) else (1, repr(d))

    dir_names: List[object] = sorted(order, key=lambda d: (-order[d], rank(d)))  # higher orders first: prefix property
    n1 = len(dir_names)
    n2 = sum(1 for d in dir_names if order[d] >= 2)
    n3 = sum(1 for d in dir_names if order[d] >= 3)
    n4 = sum(1 for d in dir_names if order[d] >= 4)
    choice = None
    for c in _INSTANTIATED:
        if c[0] >= n1 and c[1] >= n2 and c[2] >= n3 and c[3] >= n4 and (choice is None or (c[2:], c[:2]) < (choice[2:], choice[:2])):
            choice = c
    if choice is None:
        raise NotImplementedError(f"derivative set with {n1} directions / {n2} second / {n3} third / {n4} fourth-order "
                                  f"streams exceeds the instantiated kernels {_INSTANTIATED}")
    dirs_vec: List[List[float]] = []
    for d in dir_names:
        v = [0.0] * len(in_keys)
        if isinstance(d, tuple):
            v[in_keys.index(d[0])] = 1.0
            v[in_keys.index(d[1])] = float(d[2]) if len(d) > 2 else 1.0
        else:
            v[in_keys.index(d)] = 1.0
        dirs_vec.append(v)
    n1p, n2p, n3p, n4p = choice
    while len(dirs_vec) < n1p:  # padding directions are zero vectors: their streams vanish identically
        dirs_vec.append([0.0] * len(in_keys))
    streams = hp.StreamSpec(dirs_vec, n2p, n3p, n4p)
    S = streams.S
    dir_index = {d: i for i, d in enumerate(dir_names)}

    # every member carries the same stream set; its direction vectors are expressed in ITS input order (a
    # variable a member does not take contributes nothing: its derivative along it is zero), and its streams
    # are a block of rows of the constraint's U / Ubar arrays
    nets = []  # (model, StreamSpec, first U row, indices of its inputs in in_keys)
    row0: Dict[int, int] = {}
    rows = 0
    pre_nets: Dict[int, list] = {}  # id(model) -> per-feature stream expressions (input transform)
    for mm in model_list:
        feats = getattr(mm, "_traced_features", None)
        if feats:
            # a registered input transform: the network's inputs are functions phi_k of the raw variables; the
            # kernels take phi_k and its derivative streams along the directions as [S, N] blocks (EMBED_STREAMS)
            idx = None
            padded = list(dir_names) + [None] * (n1p - len(dir_names))
            blocks = []
            for k in mm.input_keys:
                phi = feats[k]
                rows_k = [phi] + [Sym.const(0.0) if d is None else directional(phi, d, 1) for d in padded]
                for kk, npk in ((2, n2p), (3, n3p), (4, n4p)):
                    rows_k += [Sym.const(0.0) if d is None else directional(phi, d, kk) for d in padded[:npk]]
                blocks.append(rows_k)
            pre_nets[id(mm)] = blocks
            spec = hp.StreamSpec([[0.0] * len(mm.input_keys) for _ in range(n1p)], n2p, n3p, n4p)
        else:
            idx = [in_keys.index(k) for k in mm.input_keys]
            spec = hp.StreamSpec([[v[j] for j in idx] for v in dirs_vec], n2p, n3p, n4p)
        nets.append((mm, spec, rows, idx))
        row0[id(mm)] = rows
        rows += len(mm.output_keys) * S
    prog = hp.Program(rows, len(in_keys))
    # inputs that are not network inputs (e.g. parameters of a boundary function) are shipped as aux arrays
    input_names = list(in_keys)
    aux_names: List[str] = []

    def aux_index(name: str) -> int:
        if name not in aux_names:
            aux_names.append(name)
        return aux_names.index(name)

    val: Dict[int, int] = {}
    param_slots = set()

    def emit_pointwise(pg: hp.Program, pnodes, pval: Dict[int, int]) -> None:
        """Inputs, aux arrays, constants and operators only (the stream programs of an input transform)."""
        for n in pnodes:
            if n.kind == "in":
                pval[id(n)] = pg.ld_in(input_names.index(n.name)) if n.name in input_names else pg.ld_aux(aux_index(n.name))
            elif n.kind == "aux":
                pval[id(n)] = pg.ld_in(input_names.index(n.name)) if n.name in input_names else pg.ld_aux(aux_index(n.name))
            elif n.kind == "const":
                pval[id(n)] = pg.const(n.value)
            elif n.kind in ("net", "param"):
                raise NotImplementedError("an input transform may only use the data-dict variables")
            elif n.op in _BINARY_OPS:
                pval[id(n)] = pg.op(_BINARY_OPS[n.op], pval[id(n.args[0])], pval[id(n.args[1])])
            else:
                pval[id(n)] = pg.op(_UNARY_OPS[n.op], pval[id(n.args[0])])

    def emit(prog: hp.Program, nodes, val: Dict[int, int]) -> None:
        """The residual program of `nodes` (in dependency order) into `prog`; val: node id -> value index."""
        for n in nodes:
            if n.kind == "in":
                if n.name in input_names:
                    val[id(n)] = prog.ld_in(input_names.index(n.name))
                else:
                    val[id(n)] = prog.ld_aux(aux_index(n.name))
            elif n.kind == "aux":
                val[id(n)] = prog.ld_in(input_names.index(n.name)) if n.name in input_names else prog.ld_aux(aux_index(n.name))
            elif n.kind == "param":
                val[id(n)] = prog.ld_param(n.comp)
                param_slots.add(n.comp)
            elif n.kind == "reduce":
                val[id(n)] = prog.ld_param(red_slot[id(n)])
            elif n.kind == "couple":  # lhs - (M v): the product is an aux column written between two launches
                val[id(n)] = prog.op(L.OP_SUB, val[id(n.args[0])], prog.ld_aux(aux_index(COUPLE_RHS_PREFIX + n.name)))
            elif n.kind == "const":
                val[id(n)] = prog.const(n.value)
            elif n.kind == "net":
                c = n.comp + row0[id(n.model)] // S  # rows of a member start at a multiple of S
                if len(n.dirs) == 0:
                    val[id(n)] = prog.ld_u(c * S)
                elif len(n.dirs) == 1:
                    val[id(n)] = prog.ld_u(c * S + 1 + dir_index[n.dirs[0]])
                else:
                    base = {2: 1 + n1p, 3: 1 + n1p + n2p, 4: 1 + n1p + n2p + n3p}[len(n.dirs)]
                    vs = sorted(set(n.dirs))
                    if len(vs) == 1:
                        val[id(n)] = prog.ld_u(c * S + base + dir_index[vs[0]])
                    elif len(n.dirs) == 2:  # polarisation: u_ab = (D2_{a+b} - D2_a - D2_b) / 2
                        a, b = sorted(n.dirs)
                        sab = prog.ld_u(c * S + base + dir_index[(a, b)])
                        sa = prog.ld_u(c * S + base + dir_index[a])
                        sb = prog.ld_u(c * S + base + dir_index[b])
                        t = prog.op(L.OP_SUB, prog.op(L.OP_SUB, sab, sa), sb)
                        val[id(n)] = prog.op(L.OP_MUL, prog.const(0.5), t)
                    elif len(n.dirs) == 3:  # u_aab = (D3_{a+b} - D3_{a-b} - 2 D3_b) / 6
                        a, b = (vs[0], vs[1]) if n.dirs.count(vs[0]) == 2 else (vs[1], vs[0])
                        lo, hi = sorted((a, b))
                        sp_ = prog.ld_u(c * S + base + dir_index[(lo, hi)])
                        sm_ = prog.ld_u(c * S + base + dir_index[(lo, hi, -1.0)])  # D3 along lo - hi = +- D3 along a - b
                        sb = prog.ld_u(c * S + base + dir_index[b])
                        t = prog.op(L.OP_SUB if a == lo else L.OP_ADD, sp_, sm_)
                        t = prog.op(L.OP_SUB, t, prog.op(L.OP_MUL, prog.const(2.0), sb))
                        val[id(n)] = prog.op(L.OP_MUL, prog.const(1.0 / 6.0), t)
                    else:  # u_aabb = (D4_{a+b} + D4_{a-b} - 2 D4_a - 2 D4_b) / 12
                        a, b = vs
                        sp_ = prog.ld_u(c * S + base + dir_index[(a, b)])
                        sm_ = prog.ld_u(c * S + base + dir_index[(a, b, -1.0)])
                        sa = prog.ld_u(c * S + base + dir_index[a])
                        sb = prog.ld_u(c * S + base + dir_index[b])
                        two = prog.const(2.0)
                        t = prog.op(L.OP_SUB, prog.op(L.OP_SUB, prog.op(L.OP_ADD, sp_, sm_), prog.op(L.OP_MUL, two, sa)),
                                    prog.op(L.OP_MUL, two, sb))
                        val[id(n)] = prog.op(L.OP_MUL, prog.const(1.0 / 12.0), t)
            else:
                if n.op in _BINARY_OPS:
                    val[id(n)] = prog.op(_BINARY_OPS[n.op], val[id(n.args[0])], val[id(n.args[1])])
                else:
                    val[id(n)] = prog.op(_UNARY_OPS[n.op], val[id(n.args[0])])


    emit(prog, nodes, val)

    loss_keys = []
    couple_rows: Dict[str, tuple] = {}  # coupling name -> (residual row, weight aux)
    periodic = []  # (residual row, label aux) -- Periodic*Loss: the label row receives the partner half's values
    causal = []  # (residual row, label aux, weight aux, area aux, causal-factor aux) -- CausalMSELoss
    for ls in losses:
        key = ls["key"]
        lab = aux_index(ls["label"]) if ls.get("label") else -1
        w = aux_index(ls["weight"]) if ls.get("weight") else -1
        ar = aux_index(ls["area"]) if ls.get("area") else -1
        if ls.get("causal"):
            # the residual's area slot carries exp(-tol * running loss) * area, written by ppsci_causal_weights
            cw = aux_index(ls["causal"])
            causal.append((len(loss_keys), lab, w, ar, cw))
            prog.n_aux = max(prog.n_aux, ar + 1)
            ar = cw
        if ls.get("periodic"):
            periodic.append((len(loss_keys), lab))
        if outputs[key].kind == "couple":
            if ls.get("kind", 0) != 0 or ls.get("causal") or ls.get("periodic") or ar >= 0 or w < 0:
                raise NotImplementedError("a batch-coupled residual is lowered under MSELoss (no area column), with its row mask "
                                          "as the weight column")
            couple_rows[outputs[key].name] = (len(loss_keys), w)
            # the term's difference d = lhs - M v - label is formed BY THE PROGRAM (the residual row then holds d, which is what
            # the transposed product of the reverse sweep needs: vbar = -M^T (2 scale w d))
            if lab >= 0:
                prog.residual(prog.op(L.OP_SUB, val[id(outputs[key])], prog.ld_aux(lab)), -1, w, ar, ls.get("scale", 1.0), 0)
                loss_keys.append(key)
                continue
        prog.residual(val[id(outputs[key])], lab, w, ar, ls.get("scale", 1.0), ls.get("kind", 0))
        loss_keys.append(key)
    for name in extra_outputs:
        prog.residual(val[id(outputs[name])], -1, -1, -1, 0.0)
        loss_keys.append(name)
    value_index = {k: val[id(v)] for k, v in outputs.items()}
    low = Lowered(model, streams, prog, input_names, aux_names, loss_keys, value_index)
    low.causal = causal
    low.periodic = periodic
    low.param_slots = sorted(param_slots)
    if reduces:
        # pass 1: the summands alone, one LINEAR term each (scale 1 / N_global for a mean): its "loss terms" ARE the reductions
        p1 = hp.Program(rows, len(in_keys))
        v1: Dict[int, int] = {}
        emit(p1, _walk([r.args[0] for r in reduces]), v1)
        coef = [1.0 / float(n_global) if r.op == "mean" else 1.0 for r in reduces]
        for r, c in zip(reduces, coef):
            p1.residual(v1[id(r.args[0])], -1, -1, -1, c, hp.LOSS_LINEAR)
        # pass 3: the residual program again + one LINEAR term per summand whose seed is c_k x dL/dR_k (a device value, written
        # by pass 2's block sums): with R held fixed, the gradient of  L(U, R) + sum_k Rbar_k c_k sum_p v_k(p)  w.r.t. U is
        # the gradient of the reference's loss, in which R depends on every point
        import copy

        p3 = copy.deepcopy(prog)
        if len(p3.res) + len(reduces) > L.MAX_RES:
            raise NotImplementedError(f"{len(p3.res)} loss terms + {len(reduces)} batch reductions exceed the {L.MAX_RES} term "
                                      "slots of one epilogue program")
        for k, (r, c) in enumerate(zip(reduces, coef)):
            p3.residual(val[id(r.args[0])], -1, -1, -1, c, hp.LOSS_LINEAR, scale_param=k + 1)
        low.reductions = {"k": len(reduces), "p1": p1, "p3": p3}
    if couples:
        import copy

        if any(c.name not in couple_rows for c in couples):
            raise NotImplementedError("a batch-coupled output needs a loss term (it has no values outside its rows)")
        # value program: v of every coupling into a row of a [C, N] buffer (scale-0 terms: values only)
        pv = hp.Program(rows, len(in_keys))
        vv: Dict[int, int] = {}
        emit(pv, _walk([c.args[1] for c in couples]), vv)
        for c in couples:
            pv.residual(vv[id(c.args[1])], -1, -1, -1, 0.0)
        # adjoint program: the residual program + one LINEAR term per coupling on v, weighted per point by
        # vbar = -M^T (2 scale w r) (an aux column written between the launches): with rhs held fixed, the gradient of
        # L(U, rhs) + sum_q vbar_q v_q  w.r.t. U is the gradient of the reference's loss, in which rhs = M v(U)
        p3 = copy.deepcopy(prog)
        if len(p3.res) + len(couples) > L.MAX_RES:
            raise NotImplementedError(f"{len(p3.res)} loss terms + {len(couples)} batch couplings exceed the {L.MAX_RES} term slots")
        items = []
        for c in couples:
            vb = aux_index(COUPLE_VBAR_PREFIX + c.name)
            p3.n_aux = max(p3.n_aux, vb + 1)
            p3.residual(val[id(c.args[1])], -1, vb, -1, 1.0, hp.LOSS_LINEAR)
            row, w = couple_rows[c.name]
            items.append({"name": c.name, "rows": c.comp[0], "cols": c.comp[1], "res_row": row, "weight_aux": w,
                          "rhs_aux": aux_names.index(COUPLE_RHS_PREFIX + c.name), "vbar_aux": vb})
        low.aux_names = aux_names
        low.couplings = {"items": items, "pv": pv, "p3": p3}
    # ---- stream programs of the input transforms: <= MAX_RES output rows per program, rows in (feature, stream) order
    pre = {}
    for mid, blocks in pre_nets.items():
        flat_rows = [e for rows_k in blocks for e in rows_k]
        progs = []
        for r0 in range(0, len(flat_rows), L.MAX_RES):
            chunk = flat_rows[r0:r0 + L.MAX_RES]
            pg = hp.Program(0, len(in_keys))
            pval: Dict[int, int] = {}
            emit_pointwise(pg, _walk(chunk), pval)
            for e in chunk:
                pg.residual(pval[id(e)], -1, -1, -1, 0.0)
            progs.append((pg, r0, len(chunk)))
        pre[mid] = progs
    low.nets = [(mm, spec, r, idx, pre.get(id(mm))) for mm, spec, r, idx in nets]
    return low
