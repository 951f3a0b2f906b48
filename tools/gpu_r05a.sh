#!/bin/bash
# round 5, first GPU pass: compile-time residual programs in the fused tile kernel (A/B against the VM), phase timers, parity
cd /root/repo
O=gpurun_out/r05a
mkdir -p $O
timeout 300 python tools/fused_bench.py 100000 4096 > $O/fused_static.json 2> $O/fused_static.err
PPSCI_STATIC_PROGRAM=0 timeout 300 python tools/fused_bench.py 100000 > $O/fused_vm.json 2> $O/fused_vm.err
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 100000 > $O/phases_static.json 2> $O/phases.err
timeout 900 python -m pytest tests/test_static_programs.py tests/test_fused_step.py tests/test_golden_bench_nets.py -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err
cat $O/fused_static.json $O/fused_vm.json
