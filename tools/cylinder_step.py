"""examples/cylinder2d_unsteady.py's iteration (the reference's TIPC case) in isolation, for rocprofv3:  python tools/cylinder_step.py"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    sys.stdout = sys.stderr
    with tempfile.TemporaryDirectory() as tmp:
        r = bench.extra_cylinder2d(tmp, 30, 5)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "published_reference")}), file=sys.__stdout__)
