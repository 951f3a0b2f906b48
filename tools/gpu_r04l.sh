#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04l
timeout 900 python -m pytest tests/test_fno.py tests/test_fno_native.py tests/test_fno_net.py tests/test_golden_fno.py tests/test_abi.py tests/test_api_pdes.py tests/test_golden_bench_nets.py -m gpu -x -q > gpurun_out/r04l/tests.log 2>&1
tail -3 gpurun_out/r04l/tests.log
timeout 300 python tools/tfno_step.py > gpurun_out/r04l/tfno.log 2>&1
grep -v amdgpu.ids gpurun_out/r04l/tfno.log | tail -5
