#!/bin/bash
# determinism of the fused step after the float4 rewrite of its pointwise arithmetic + the fused-kernel tests + the main kernel's time
cd /root/repo
export PYTHONPATH=.
PROBE_RUNS=24 timeout 600 python tools/det_probe4.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_fused_step.py tests/test_static_programs.py tests/test_golden_bench_nets.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/fused_main_time.py 100000; PPSCI_STATIC_PROGRAM=0 timeout 200 python tools/fused_main_time.py 100000; done
