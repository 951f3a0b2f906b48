#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_fused_step.py tests/test_kernels.py tests/test_one_launch.py tests/test_golden_bench_nets.py -m gpu -x -q > gpurun_out/r04f/tests.log 2>&1
timeout 300 python tools/fused_bench.py 100000 4096 16384 > gpurun_out/r04f/fused_bench.log 2>&1
tail -2 gpurun_out/r04f/tests.log
cat gpurun_out/r04f/fused_bench.log | grep net
