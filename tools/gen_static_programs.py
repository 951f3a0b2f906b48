"""Generates paddlescience_amd/csrc/epi_static_programs.h: the residual programs of the equation classes the reference
ships (and of the hand-built programs bench.py / the tests time) as COMPILE-TIME tables for the fused tile kernel and the
one-launch step kernel (csrc/epi_static.h).

A residual program reaches the kernels as data (ppsci_epilogue_desc) and is normally run by the epilogue VM: on the
16 point-lanes of one wave, ~165-220 cycles per step, with the workgroup's other waves parked behind it (20 % of a tile
of the fused kernel).  The same table known at COMPILE time unrolls into a few dozen straight-line VALU instructions that
every wave evaluates redundantly for its own lanes -- no serial wave, no second barrier, no parking of the stash.  The
plan matches a program against these tables word for word (structure only: the constants' VALUES stay run-time data, so
one table serves AllenCahn(eps) for every eps); a program that matches none runs on the VM as before.

Every table below is produced by running the SAME lowering the API path uses (compile.trace_exprs + graph.lower on the
package's equation classes) or the same hand-built hp.Program the bench / tests use, then pre-decoding it exactly as
csrc/epilogue_vm.h epi_fast_encode does (tests/test_static_programs.py checks this port against the C function and
that the committed header is what this script generates).

    python tools/gen_static_programs.py            # rewrite the header
    python tools/gen_static_programs.py --check    # exit 1 if the committed header differs

Reference residuals: /root/reference/ppsci/equation/pde/allen_cahn.py:56-64, laplace.py:40-55, poisson.py:40-56,
navier_stokes.py:96-160, helmholtz.py; loss: /root/reference/ppsci/loss/mse.py:82-105.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "paddlescience_amd", "csrc", "epi_static_programs.h")
MAX_AUX_REGS = 4  # aux arrays a static program may read (prefetched into registers per tile)


def predecode(e):
    """Python port of csrc/epilogue_vm.h epi_fast_encode: (steps, loads, terms, n_consts) or None."""
    from paddlescience_amd import _lib as L

    if e.n_instr > 127:
        return None
    loads = []
    for i in range(e.n_instr):
        ins = e.prog[i]
        kind = {L.OP_LD_IN: 0, L.OP_LD_U: 1, L.OP_LD_AUX: 2, L.OP_CONST: 3}.get(ins.op, -1)
        if kind < 0:
            continue
        if len(loads) == 64 or ins.a > 127 or ins.a < 0:
            return None
        loads.append(i | ((0 if kind == 3 else ins.a) << 7) | (kind << 14))
    terms = []
    for k in range(e.n_res):
        r = e.res[k]
        if r.label > 30 or r.weight > 30 or r.area > 30 or r.kind != 0:
            return None
        terms.append(r.value | ((r.label + 1) << 7) | ((r.weight + 1) << 12) | ((r.area + 1) << 17))

    def word(a, b, i, prod, sx, sy, rx, ry):
        return a | (b << 7) | (i << 14) | (prod << 21) | (sx << 22) | (sy << 24) | (rx << 26) | (ry << 28) | ((1 if a == b else 0) << 30)

    steps = []
    for i in range(e.n_instr):
        ins = e.prog[i]
        if ins.op in (L.OP_LD_IN, L.OP_LD_U, L.OP_LD_AUX, L.OP_CONST):
            continue
        if ins.op == L.OP_ADD:
            w = word(ins.a, ins.b, i, 0, 1, 1, 1, 1)
        elif ins.op == L.OP_SUB:
            w = word(ins.a, ins.b, i, 0, 1, 2, 1, 2)
        elif ins.op == L.OP_MUL:
            w = word(ins.a, ins.b, i, 1, 0, 0, 0, 0)
        elif ins.op == L.OP_NEG:
            w = word(ins.a, ins.a, i, 0, 2, 0, 2, 0)
        elif ins.op == L.OP_DETACH:
            w = word(ins.a, ins.a, i, 0, 1, 0, 0, 0)
        else:
            return None
        if len(steps) == 64:
            return None
        steps.append(w)
    return steps, loads, terms


# ---------------------------------------------------------------------------------------------- the programs
def _lower(model, exprs, input_keys, label_keys, weight_keys=()):
    """What compile.CompiledConstraint does for an MSE constraint, without device buffers."""
    import paddlescience_amd as ppsci
    from paddlescience_amd import compile as cp
    from paddlescience_amd import graph
    from paddlescience_amd.compile import LABEL_PREFIX, WEIGHT_PREFIX

    loss = ppsci.loss.MSELoss("mean")
    outputs = cp.trace_exprs(model, input_keys, exprs, (), None, [])
    for k in label_keys:
        if k not in outputs:
            outputs[k] = graph.Sym.net(model, model.output_keys.index(k))
    losses = [dict(key=k, label=LABEL_PREFIX + k, weight=(WEIGHT_PREFIX + k) if k in weight_keys else None, area=None,
                   scale=loss.term_scale(k, 1000), kind=0, causal=None, periodic=False) for k in label_keys]
    low = graph.lower(outputs, losses, ())
    return low.program.build(), low.streams


def cases():
    """(name, citation / origin, EpilogueDesc, n1, n2, d_out)."""
    import paddlescience_amd as ppsci
    from paddlescience_amd import device

    device.set_device("cpu")  # parameters of the throw-away models below live on the host
    import bench
    from tests import test_one_launch as T1

    out = []
    e = bench.allen_cahn_program(1000)
    out.append(("allen_cahn_handbuilt", "bench.py allen_cahn_program / tests `allen_cahn` (allen_cahn.py:56-64)", e, 2, 1, 1))
    for kind in ("laplace", "value"):
        e, st, _ = T1._program(kind, 1000)
        out.append((f"test_{kind}", f"tests/test_one_launch.py _program('{kind}')", e, len(st.dirs), st.n2, 1))
    # the API path: equation classes lowered exactly as a constraint does
    m_tx = ppsci.arch.MLP(("t", "x"), ("u",), 2, 16, "tanh")
    m_xy = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16, "tanh")
    m_xyz = ppsci.arch.MLP(("x", "y", "z"), ("u",), 2, 16, "tanh")
    m_p = ppsci.arch.MLP(("x", "y"), ("p",), 2, 16, "tanh")
    m_uvp = ppsci.arch.MLP(("x", "y"), ("u", "v", "p"), 2, 16, "tanh")
    m_tuvp = ppsci.arch.MLP(("t", "x", "y"), ("u", "v", "p"), 2, 16, "tanh")

    def api(name, cite, model, exprs, keys, labels, weights=()):
        e, st = _lower(model, exprs, keys, labels, weights)
        out.append((name, cite, e, len(st.dirs), st.n2, len(model.output_keys)))

    ac = ppsci.equation.AllenCahn(0.01 ** 2)
    api("allen_cahn", "ppsci.equation.AllenCahn (allen_cahn.py:56-64), MSE against a label", m_tx, ac.equations, ("t", "x"), ["allen_cahn"])
    api("allen_cahn_w", "AllenCahn with per-point weights", m_tx, ac.equations, ("t", "x"), ["allen_cahn"], ["allen_cahn"])
    lap = ppsci.equation.Laplace(2)
    api("laplace2d", "ppsci.equation.Laplace(dim=2) (laplace.py:40-55)", m_xy, lap.equations, ("x", "y"), ["laplace"])
    api("laplace2d_w", "Laplace(dim=2) with per-point weights", m_xy, lap.equations, ("x", "y"), ["laplace"], ["laplace"])
    lap3 = ppsci.equation.Laplace(3)
    api("laplace3d", "ppsci.equation.Laplace(dim=3)", m_xyz, lap3.equations, ("x", "y", "z"), ["laplace"])
    poi = ppsci.equation.Poisson(2)
    api("poisson2d", "ppsci.equation.Poisson(dim=2) (poisson.py:40-56)", m_p, poi.equations, ("x", "y"), ["poisson"])
    api("value_u", "boundary / initial / supervised constraint on one output: u against a label", m_xy, {"u": lambda d: d["u"]},
        ("x", "y"), ["u"])
    api("value_u_w", "the same with per-point weights", m_xy, {"u": lambda d: d["u"]}, ("x", "y"), ["u"], ["u"])
    api("value_u_tx", "u(t, x) against a label (initial / boundary condition of Allen-Cahn)", m_tx, {"u": lambda d: d["u"]},
        ("t", "x"), ["u"])
    ns = ppsci.equation.NavierStokes(0.01, 1.0, 2, False)
    api("navier_stokes2d", "ppsci.equation.NavierStokes(nu, rho, dim=2, time=False) (navier_stokes.py:96-160)", m_uvp, ns.equations,
        ("x", "y"), ["continuity", "momentum_x", "momentum_y"])
    api("value_uv", "LDC walls: u, v against labels", m_uvp, {"u": lambda d: d["u"], "v": lambda d: d["v"]}, ("x", "y"), ["u", "v"])
    nst = ppsci.equation.NavierStokes(0.01, 1.0, 2, True)
    api("navier_stokes2d_t", "NavierStokes(nu, rho, dim=2, time=True)", m_tuvp, nst.equations, ("t", "x", "y"),
        ["continuity", "momentum_x", "momentum_y"])
    return out


# ---------------------------------------------------------------------------------------------- the header
def emit(cs) -> str:
    lines = ["// epi_static_programs.h -- GENERATED by tools/gen_static_programs.py (python tools/gen_static_programs.py); do not edit.",
             "// Residual programs known at compile time (see epi_static.h): tables in the encoding of epi_fast_encode",
             "// (epilogue_vm.h), one struct per program.  Structure only -- the constants' values are run-time data.",
             "#pragma once", ""]
    names, seen = [], {}
    pid = 0
    for name, cite, e, n1, n2, m in cs:
        dec = predecode(e)
        S = 1 + n1 + n2
        if dec is None:
            print(f"  skipped {name}: not a +,-,*,neg,detach program under MSE terms", file=sys.stderr)
            continue
        steps, loads, terms = dec
        if e.n_aux > MAX_AUX_REGS or e.n_res < 1:
            print(f"  skipped {name}: {e.n_aux} aux arrays", file=sys.stderr)
            continue
        key = (tuple(steps), tuple(loads), tuple(terms), e.n_instr, n1, n2, m, e.n_in, e.n_aux)
        if key in seen:
            print(f"  {name}: same table as {seen[key]}", file=sys.stderr)
            continue
        seen[key] = name
        pid += 1
        names.append(f"EpiProg_{name}")

        def arr(fn, vals):
            body = ", ".join(f"0x{v:08x}u" for v in vals) if vals else "0u"
            return (f"  __host__ __device__ static constexpr unsigned {fn}(int k) {{\n"
                    f"    constexpr unsigned t[] = {{{body}}};\n    return t[k];\n  }}")

        lines += [f"// {cite}",
                  f"struct EpiProg_{name} {{",
                  f"  static constexpr int ID = {pid}, N1 = {n1}, N2 = {n2}, M = {m}, S = {S}, MS = {m * S};",
                  f"  static constexpr int NI = {e.n_instr}, NL = {len(loads)}, NS = {len(steps)}, NR = {e.n_res}, NIN = {e.n_in}, NAUX = {e.n_aux};",
                  f"  static constexpr const char* name() {{ return \"{name}\"; }}",
                  arr("load", loads), arr("step", steps), arr("term", terms), "};", ""]
    lines += [f"#define EPI_STATIC_MAX_AUX {MAX_AUX_REGS}",
              "using EpiStaticPrograms = epi_typelist<" + ", ".join(names) + ">;", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    txt = emit(cases())
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == txt
        print("up to date" if ok else "STALE: run python tools/gen_static_programs.py")
        sys.exit(0 if ok else 1)
    with open(OUT, "w") as f:
        f.write(txt)
    print(f"wrote {OUT}")
