#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04k
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04k/tests.log 2>&1
tail -3 gpurun_out/r04k/tests.log
timeout 600 python tools/one_launch_bench.py > gpurun_out/r04k/one_launch.log 2>&1
grep -v amdgpu.ids gpurun_out/r04k/one_launch.log | head -12
