#!/bin/bash
# UNO on the GPU box: its tests, the step at the reference config (16 x 16 and 64 x 64), the example.
cd /root/repo
O=gpurun_out/uno
mkdir -p $O
timeout 900 python -m pytest tests/test_uno.py tests/test_fno_native.py -m gpu -q 2>&1 | tail -5
timeout 300 python tools/uno_step.py 30 16 2>&1 | tail -2
timeout 300 python tools/uno_step.py 30 64 2>&1 | tail -2
PPSCI_HIP_GRAPH=0 timeout 300 python tools/uno_step.py 30 16 2>&1 | tail -1
timeout 600 python examples/uno_darcyflow.py epochs=4 output_dir=/tmp/uno_out data_dir=/tmp/uno_data 2>&1 | grep -E "Eval|Error|error" | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o uno -- python /root/repo/tools/uno_step.py 30 16 > /root/repo/$O/prof.log 2>&1
cd /root/repo
find gpurun_out/uno -name "*_kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/uno/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernels", len(rows), "total_ms_per_step", tot / 60 / 1e6, "launches_per_step", sum(int(r["Calls"]) for r in rows) / 60)
    for r in rows[:14]:
        print(f'{float(r["TotalDurationNs"])/60/1e3:8.1f} us/step  {int(r["Calls"])/60:5.1f}  {float(r["AverageNs"])/1e3:7.1f} us  {r["Name"][:90]}')
PY
