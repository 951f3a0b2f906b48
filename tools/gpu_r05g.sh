#!/bin/bash
# round 5: collectives in the C ABI (one rank), the Darcy end-to-end test on the device, FNO side-stream A/B
cd /root/repo
O=gpurun_out/r05g
mkdir -p $O
timeout 1200 python -m pytest tests/test_comm.py tests/test_neuralop_data.py tests/test_fno_net.py tests/test_fno_native.py tests/test_golden_fno.py tests/test_fullsize.py -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
for i in 1 2; do
  timeout 300 python tools/tfno_step.py 60 >> $O/tfno_side.txt 2>> $O/tfno.err
  PPSCI_FNO_SIDE_STREAM=0 timeout 300 python tools/tfno_step.py 60 >> $O/tfno_noside.txt 2>> $O/tfno.err
done
echo side; cat $O/tfno_side.txt; echo no side; cat $O/tfno_noside.txt
