"""Allen-Cahn at the reference yaml's shape (4 x 256 tanh, periods {x: 2.0}, 100 000 points) step in isolation, for rocprofv3:
python tools/ac256_step.py [steps]"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PPSCI_BENCH_PURE_STEPS"] = "1"  # nothing but the training steps (launches / steps = launches per step)
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    with tempfile.TemporaryDirectory() as tmp:
        r = bench.secondary_ac256(tmp, steps, 5)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step")}))
