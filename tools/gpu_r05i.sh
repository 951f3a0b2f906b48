#!/bin/bash
# round 5, after the float4 rewrite of the fused kernel's pointwise arithmetic: whole GPU suite, driver-style bench line,
# primary-config profile (kernel trace + PMC passes), smoke
cd /root/repo
O=gpurun_out/r05i
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05i/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"].get("mfma_busy"), r["parity"])
print({e["config"][:28]: round(e["ms_per_step"], 4) for e in r.get("secondary", []) if isinstance(e, dict) and "ms_per_step" in e})
PY
bash tools/profile_bench.sh r05_bench > $O/profile.log 2>&1
tail -3 $O/profile.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
