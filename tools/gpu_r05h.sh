#!/bin/bash
# round 5: the whole GPU suite + the driver-style bench line on the current build
cd /root/repo
O=gpurun_out/r05h
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05h/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"].get("mfma_busy"), r["parity"])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
