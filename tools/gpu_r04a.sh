#!/bin/bash
# first GPU pass of round 4: fused tile kernel correctness + timings + profile
cd /root/repo
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_fused_step.py tests/test_one_launch.py tests/test_kernels.py tests/test_golden_bench_nets.py -m gpu -x -q > gpurun_out/r04a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04a/tests.log
timeout 600 python tools/fused_bench.py > gpurun_out/r04a/fused_bench.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-strong > gpurun_out/r04a/bench_primary.json 2> gpurun_out/r04a/bench_primary.err
PPSCI_ONE_LAUNCH=0 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-strong > gpurun_out/r04a/bench_primary_sep.json 2> gpurun_out/r04a/bench_primary_sep.err
timeout 1500 bash tools/profile_bench.sh r04a_bench > gpurun_out/r04a/profile.log 2>&1
tail -3 gpurun_out/r04a/tests.log; cat gpurun_out/r04a/fused_bench.log; cut -c1-600 gpurun_out/r04a/bench_primary.json
