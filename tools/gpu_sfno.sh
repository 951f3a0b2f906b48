#!/bin/bash
# SFNO on the GPU box: its tests, the step at the reference config (32 x 64 and 64 x 128), a kernel trace.
cd /root/repo
O=gpurun_out/sfno
mkdir -p $O
timeout 900 python -m pytest tests/test_sfno.py tests/test_fno_native.py tests/test_uno.py tests/test_fullsize.py -m gpu -q -k "sfno or fno or uno" 2>&1 | tail -3
timeout 300 python tools/sfno_step.py 30 32 64 2>&1 | tail -1
timeout 300 python tools/sfno_step.py 30 64 128 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o sfno -- python /root/repo/tools/sfno_step.py 30 32 64 > /root/repo/$O/prof.log 2>&1
cd /root/repo
find gpurun_out/sfno -name "*_kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sfno/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total_us_per_step", tot / 60 / 1e3, "launches_per_step", sum(int(r["Calls"]) for r in rows) / 60)
    for r in rows[:12]:
        print(f'{float(r["TotalDurationNs"])/60/1e3:8.1f} us/step  {int(r["Calls"])/60:5.1f}x  {float(r["AverageNs"])/1e3:7.1f} us  {r["Name"][:80]}')
PY
