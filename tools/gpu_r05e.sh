#!/bin/bash
# round 5, fifth GPU pass: layer-by-layer XDL reverse kernel for padded width 256 (AC 4 x 256), tests, the whole bench line
cd /root/repo
O=gpurun_out/r05e
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels.py tests/test_golden_bench_nets.py -m gpu -x -q -k "layerwise or wide or bench_nets" > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 300 python tools/wide_bench.py > $O/wide.jsonl 2> $O/wide.err
cat $O/wide.jsonl
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05e/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["parity"])
for e in r.get("secondary", []):
    print({k: e.get(k) for k in ("config", "ms_per_step", "error", "parity", "reverse_ms_round2_kernel")}, (e.get("roofline") or {}).get("kernel_ms"), (e.get("roofline") or {}).get("frac"))
PY
