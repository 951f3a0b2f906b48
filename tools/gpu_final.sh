#!/bin/bash
# Final pass of a round on the GPU box: whole -m gpu suite, the driver-style bench line (plain and under torchrun with one rank),
# smoke.  Outputs under gpurun_out/final/.
cd /root/repo
O=gpurun_out/final
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
tail -2 $O/tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/final/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["kernel_ms"], round(r["roofline"]["frac"], 3), r["parity"]["grad_rel_l2"])
print({e["config"][:28]: round(e["ms_per_step"], 4) for e in r.get("secondary", []) if isinstance(e, dict) and "ms_per_step" in e})
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-strong > $O/bench_torchrun.json 2> $O/bench_torchrun.err
python -c "import json; r = json.load(open('gpurun_out/final/bench_torchrun.json')); print('torchrun 1 rank:', r['value'], r['ms_per_step'], r['n_gpus'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
