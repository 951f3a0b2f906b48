#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04q
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04q/tests.log 2>&1
tail -3 gpurun_out/r04q/tests.log
timeout 1200 python bench.py > gpurun_out/r04q/bench.json 2> gpurun_out/r04q/bench.err
tail -c 600 gpurun_out/r04q/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04q/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','roofline')})
for k,v in d.items():
    if isinstance(v,dict) and 'ms_per_step' in v: print(k, v.get('ms_per_step'), v.get('ms_per_step_hip_events'))
for e in d.get('secondary',[]) if isinstance(d.get('secondary'),list) else []:
    print(e.get('config','')[:40], e.get('ms_per_step'), e.get('ms_per_step_hip_events'))
PY
