#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04m
timeout 900 python -m pytest tests/test_fno.py tests/test_fno_native.py tests/test_fno_net.py tests/test_golden_fno.py tests/test_abi.py -m gpu -x -q > gpurun_out/r04m/tests.log 2>&1
tail -3 gpurun_out/r04m/tests.log
timeout 300 python tools/tfno_step.py 50 > gpurun_out/r04m/tfno.log 2>&1
grep -v amdgpu.ids gpurun_out/r04m/tfno.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04m/prof -o tfno -- python /root/repo/tools/tfno_step.py 30 > /root/repo/gpurun_out/r04m/prof.log 2>&1
cd /root/repo
python tools/tfno_timeline.py gpurun_out/r04m/prof/tfno_results.db > gpurun_out/r04m/timeline.txt; tail -22 gpurun_out/r04m/timeline.txt
