"""Fused tile kernel (csrc/taylor_fused.inc) against the separate launches on Allen-Cahn-type steps (HIP events on the launch
stream; one JSON line per configuration):  python tools/fused_bench.py [points ...]
  sep_us          forward + epilogue + reverse + reductions + Adam as separate launches
  fused_tree_us   weight split + fused kernel ending in the reduction tree (gradient, loss, Adam in the launch)
  fused_ext_us    weight split + fused kernel + the two reduction kernels + loss sum + Adam
  fused_hyb_us    weight split + fused kernel with the FIRST tree level inside + one reduction kernel (measured slower than fused_ext)
  main_us         the fused kernel alone (no weight split, no reduction)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from paddlescience_amd import _lib as L  # noqa: E402
from paddlescience_amd import hotpath as hp  # noqa: E402
from paddlescience_amd.engine import Engine, FusedConstraint  # noqa: E402


def run(n, hidden=4, width=64, reps=30):
    dev = torch.device("cuda", 0)
    flat = bench.bench_weights(2, [width] * hidden, 1)
    X = np.random.default_rng(42).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
    out = {"net": f"{hidden}x{width}", "points": n}
    grads = {}
    for tag, one, tail in (("sep", False, -1), ("fused_tree", True, 0), ("fused_ext", True, 1), ("fused_hyb", True, 2)):
        L.lib().ppsci_set_step_tail(tail)
        L.lib().ppsci_set_static_program(0 if os.environ.get("PPSCI_STATIC_PROGRAM", "1") == "0" else 1)  # A/B: the VM
        lay = hp.NetLayout(2, hidden, width, 1, "tanh")
        xs = [torch.tensor(X[:, j].copy(), device=dev) for j in range(2)]
        cst = FusedConstraint("EQ", lay, hp.StreamSpec([[0.0, 1.0], [1.0, 0.0]], 1), bench.allen_cahn_program(n), xs, [],
                              ["allen_cahn"])
        eng = Engine(lay, torch.tensor(flat, device=dev))
        eng.one_launch = one
        eng.forward_backward([cst])
        torch.cuda.synchronize()
        grads[tag] = eng.grad.cpu().numpy().copy()
        for _ in range(5):
            eng.train_step([cst], 1e-3)
        out[tag + "_us"] = round(bench.time_events(lambda: eng.train_step([cst], 1e-3), reps) * 1e6, 2)
        if one:
            assert cst._step_kind == hp.STEP_FUSED_TILE
            out["static_program"] = cst._step_plan.static_program
            out["main_us"] = round(bench.time_events(cst._step_plan.run_main, reps) * 1e6, 2)
    L.lib().ppsci_set_step_tail(-1)
    out["grad_rel_tree_vs_sep"] = bench.rel(grads["fused_tree"], grads["sep"])
    out["grad_rel_ext_vs_sep"] = bench.rel(grads["fused_ext"], grads["sep"])
    out["grad_rel_hyb_vs_sep"] = bench.rel(grads["fused_hyb"], grads["sep"])
    p_mat = 2 * width + (hidden - 1) * width * width + width
    out["main_tflops"] = round(6.0 * p_mat * 4 * n / (out["main_us"] * 1e-6) / 1e12, 1)
    return out


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [100_000, 4096, 16_384, 1_000_000]
    for n in sizes:
        print(json.dumps(run(n)), flush=True)
