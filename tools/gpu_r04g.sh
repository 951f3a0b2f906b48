#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04g/tests.log 2>&1
tail -4 gpurun_out/r04g/tests.log
timeout 900 python bench.py > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r04g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'].get(k) for k in ('kernel','achieved','frac','pipe_frac','mfma_busy','kernel_ms')})
print(d['parity']); print(d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_best_thread'))
for e in d.get('secondary',[]):
    print({k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ('config','value','ms_per_step','error','parity')} if 'error' not in e else e)
print(d.get('strong_scaling'))
PY
tail -5 gpurun_out/r04g/bench.err
