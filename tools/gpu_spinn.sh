#!/bin/bash
# SPINN (cfg 5) on the GPU box: tests, step time per setting of the tile knob (1 reverse only = default, 0 per point, 2 both), profile.
cd /root/repo
O=gpurun_out/spinn
mkdir -p $O
rm -f $O/*.jsonl
timeout 900 python -m pytest tests/test_spinn.py tests/test_golden_spinn.py tests/test_abi.py -m gpu -q -x > $O/tests.log 2>&1
tail -3 $O/tests.log
for i in 1 2; do
  for m in 1 0 2; do PPSCI_MODMLP_TILE=$m timeout 300 python tools/spinn_step.py 300 >> $O/mode$m.jsonl 2>> $O/err.log; done
done
for m in 1 0 2; do echo mode $m; cat $O/mode$m.jsonl; done
bash tools/profile_bench.sh r06_spinn python /root/repo/tools/spinn_step.py 50 > $O/profile.log 2>&1
python tools/summarize_profile.py r06_spinn r06_spinn > /dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('profiles/r06_spinn_kernel_stats.csv')):
    if 'at::native' in r['Name'] or 'rocclr' in r['Name']: continue
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3, 2), r['Percentage'])
PY
cp profiles/r06_spinn_* $O/ 2>/dev/null
