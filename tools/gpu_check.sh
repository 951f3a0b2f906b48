#!/bin/bash
# A development pass on the GPU box (through gpurun): the -m gpu suite (or the files given as arguments), the driver-style
# bench line, smoke.  Outputs under gpurun_out/check/.
cd /root/repo
O=gpurun_out/check
mkdir -p $O
if [ $# -gt 0 ]; then T="$*"; else T="tests"; fi
timeout 2400 python -m pytest $T -m gpu -q -x > $O/tests.log 2>&1
tail -4 $O/tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -c 600 $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/check/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "windows")})
print("kernel_ms", r["roofline"]["kernel_ms"], "frac", round(r["roofline"]["frac"], 3), "parity", r["parity"])
print({e["config"][:28]: round(e["ms_per_step"], 4) for e in r.get("secondary", []) if isinstance(e, dict) and "ms_per_step" in e})
print("cpu_baseline", {k: v for k, v in r.get("cpu_baseline", {}).items() if k != "sample"})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
