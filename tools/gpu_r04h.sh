#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_fused_step.py tests/test_fullsize.py -m gpu -x -q > gpurun_out/r04h/tests.log 2>&1
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 100000 > gpurun_out/r04h/phases_100k.json 2> gpurun_out/r04h/phases.err
timeout 300 python tools/fused_bench.py 100000 4096 16384 1000000 > gpurun_out/r04h/fused_bench.log 2>&1
tail -3 gpurun_out/r04h/tests.log
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r04h/phases_100k.json'))
print(d['main_us'], d['sum_wave0'])
for k,v in d['wave0_p10_p90'].items(): print(k, v)
print(d['program_inner_cycles_per_tile'])
PY
cat gpurun_out/r04h/fused_bench.log | grep net; tail -3 gpurun_out/r04h/phases.err
