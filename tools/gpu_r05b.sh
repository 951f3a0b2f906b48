#!/bin/bash
# round 5, second GPU pass: one-kernel tail (mode 3) against the two reduction kernels (mode 1), ablations of the fused kernel, tests
cd /root/repo
O=gpurun_out/r05b
mkdir -p $O
for t in 3 1 3 1; do PPSCI_STEP_TAIL=$t timeout 200 python tools/fused_main_time.py 100000 >> $O/tail.jsonl 2>> $O/tail.err; done
cat $O/tail.jsonl
bash tools/fused_ablate.sh run $O > /dev/null
cat $O/ablate.jsonl
timeout 900 python -m pytest tests/test_static_programs.py tests/test_fused_step.py -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
