#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04d
timeout 600 python -m pytest tests/test_fused_step.py -m gpu -x -q > gpurun_out/r04d/tests.log 2>&1
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 100000 > gpurun_out/r04d/phases_100k.json 2> gpurun_out/r04d/phases.err
timeout 300 python tools/fused_bench.py 100000 4096 16384 > gpurun_out/r04d/fused_bench.log 2>&1
tail -2 gpurun_out/r04d/tests.log; python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04d/phases_100k.json'))
print(d['main_us'], d['sum_wave0'])
for k,v in d['wave0_p10_p90'].items(): print(k, v)
PY
cat gpurun_out/r04d/fused_bench.log; tail -3 gpurun_out/r04d/phases.err
