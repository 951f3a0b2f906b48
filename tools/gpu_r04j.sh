#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04j
timeout 600 python -m pytest tests/test_fused_step.py -m gpu -x -q > gpurun_out/r04j/tests.log 2>&1
timeout 900 bash tools/profile_bench.sh r04_bench > gpurun_out/r04j/profile.log 2>&1
timeout 300 python tools/fused_bench.py 100000 > gpurun_out/r04j/fused_bench.log 2>&1
tail -2 gpurun_out/r04j/tests.log; grep net gpurun_out/r04j/fused_bench.log
