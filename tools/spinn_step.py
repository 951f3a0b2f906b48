"""cfg5 (SPINN Helmholtz3D, 128^3 grid) step in isolation, for rocprofv3:  python tools/spinn_step.py [steps]"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PPSCI_BENCH_PURE_STEPS"] = "1"  # nothing but the training steps (launches / steps = launches per step)
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    if os.environ.get("PPSCI_MODMLP_TILE", "1") != "1":  # A/B: 0 the per-point kernels of the branch nets, 2 tiles in both sweeps
        from paddlescience_amd import _lib as L

        L.lib().ppsci_set_modmlp_tile(int(os.environ["PPSCI_MODMLP_TILE"]))
    with tempfile.TemporaryDirectory() as tmp:
        r = bench.secondary_spinn(tmp, steps, 10)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "parity", "roofline") if k in r}))
