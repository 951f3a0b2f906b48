#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_fused_step.py -m gpu -x -q > gpurun_out/r04c/tests.log 2>&1
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 100000 > gpurun_out/r04c/phases_100k.json 2> gpurun_out/r04c/phases.err
timeout 300 python tools/fused_bench.py 100000 4096 16384 > gpurun_out/r04c/fused_bench.log 2>&1
tail -3 gpurun_out/r04c/tests.log; cat gpurun_out/r04c/phases_100k.json gpurun_out/r04c/fused_bench.log; tail -3 gpurun_out/r04c/phases.err
