#!/bin/bash
# round 5, third GPU pass: dot2 split + forward-plane reuse + raw layer 0 in the STATIC kernels
cd /root/repo
O=gpurun_out/r05d
mkdir -p $O
for i in 1 2; do
  timeout 200 python tools/fused_main_time.py 100000 >> $O/main.jsonl 2>> $O/main.err
  PPSCI_STATIC_PROGRAM=0 timeout 200 python tools/fused_main_time.py 100000 >> $O/main.jsonl 2>> $O/main.err
done
cat $O/main.jsonl
bash tools/fused_ablate.sh run $O > /dev/null
cat $O/ablate.jsonl
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 100000 > $O/phases_static.json 2> $O/phases.err
timeout 900 python -m pytest tests/test_static_programs.py tests/test_fused_step.py tests/test_golden_bench_nets.py tests/test_fno_net.py -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05d/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["parity"])
PY
