#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/tfno_profile
timeout 900 python -m pytest tests/test_fno.py tests/test_fno_native.py tests/test_fno_net.py tests/test_golden_fno.py tests/test_abi.py tests/test_fullsize.py -m gpu -x -q > gpurun_out/tfno_profile/tests.log 2>&1
tail -3 gpurun_out/tfno_profile/tests.log
for npx in 0; do
  PPSCI_PW_NPX=$npx timeout 300 python tools/tfno_step.py 50 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/npx=$npx /" | tee -a gpurun_out/tfno_profile/tfno.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/tfno_profile/prof -o tfno -- python /root/repo/tools/tfno_step.py 30 > /root/repo/gpurun_out/tfno_profile/prof.log 2>&1
cd /root/repo
python tools/tfno_timeline.py gpurun_out/tfno_profile/prof/tfno_results.db > gpurun_out/tfno_profile/timeline.txt; tail -20 gpurun_out/tfno_profile/timeline.txt
