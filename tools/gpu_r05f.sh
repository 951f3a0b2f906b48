#!/bin/bash
# round 5: rocprofv3 profiles of every config on the current build (kernel trace + PMC passes, each in its own run) and the bench line
cd /root/repo
O=gpurun_out/r05f
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05f/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["parity"])
PY
bash tools/profile_all.sh r05 > $O/profile.log 2>&1
tail -3 $O/profile.log
