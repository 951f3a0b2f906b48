#!/bin/bash
# The fused reduce + Adam tail (SPINN, FNO) on the GPU box: tests, A/B of the step times.
cd /root/repo
O=gpurun_out/tail
mkdir -p $O; rm -f $O/*.txt
timeout 1200 python -m pytest tests/test_reduce_adam.py tests/test_spinn.py tests/test_fno_net.py tests/test_fno_native.py tests/test_golden_fno.py tests/test_neuralop_data.py tests/test_abi.py -m gpu -q -x > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -1
for i in 1 2; do
  for f in 1 0; do
    PPSCI_FUSED_REDUCE_ADAM=$f timeout 300 python tools/spinn_step.py 300 >> $O/spinn_$f.txt 2>> $O/err.log
    PPSCI_FUSED_REDUCE_ADAM=$f timeout 300 python tools/tfno_step.py 200 2>> $O/err.log | tail -1 >> $O/tfno_$f.txt
  done
done
for f in 1 0; do echo "fused=$f"; cat $O/spinn_$f.txt $O/tfno_$f.txt; done
for i in 1 2; do PPSCI_FNO_FUSE_CONTRACT=0 timeout 300 python tools/tfno_step.py 200 2>> $O/err.log | tail -1 >> $O/tfno_nocontract.txt; done
echo "PPSCI_FNO_FUSE_CONTRACT=0 (the forward contraction as its own launch)"; cat $O/tfno_nocontract.txt
for i in 1 2; do PPSCI_SPINN_PARTS=1 timeout 300 python tools/spinn_step.py 300 >> $O/spinn_parts.txt 2>> $O/err.log; done
echo "parts=1 (the branch kernel sums the grid partials: slower at this size)"; cat $O/spinn_parts.txt
bash tools/profile_bench.sh r06_spinn python /root/repo/tools/spinn_step.py 50 > $O/profile.log 2>&1
find gpurun_out/prof_r06_spinn -name "*_kernel_trace.csv" -delete
