#!/bin/bash
# Final pass of a round on the GPU box: whole -m gpu suite, the driver-style bench line (plain and under torchrun with one rank),
# smoke, then rocprofv3 profiles of every config (kernel trace + PMC passes, each in its own run).  Outputs under gpurun_out/.
#   bash tools/gpu_final.sh [round tag, default r06]
cd /root/repo
R=${1:-r06}
O=gpurun_out/final
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/final/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "windows")})
print("kernel_ms", r["roofline"]["kernel_ms"], "frac", round(r["roofline"]["frac"], 3), "pipe_frac", round(r["roofline"]["pipe_frac"], 3),
      "parity", {k: v for k, v in r["parity"].items() if k != "reference"})
print({e["config"][:28]: round(e["ms_per_step"], 4) for e in r.get("secondary", []) if isinstance(e, dict) and "ms_per_step" in e})
print("cpu_baseline", {k: v for k, v in r.get("cpu_baseline", {}).items() if k != "sample"})
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-strong > $O/bench_torchrun.json 2> $O/bench_torchrun.err
python -c "import json; r = json.load(open('gpurun_out/final/bench_torchrun.json')); print('torchrun 1 rank:', r['value'], r['ms_per_step'], r['n_gpus'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_all.sh $R > $O/profile.log 2>&1
# keep what tools/summarize_profile.py reads (the stats CSV and the counter CSVs), drop the per-dispatch traces (64 MiB merge limit)
find gpurun_out -name "*_kernel_trace.csv" -delete
du -sh gpurun_out | tail -1
