#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04b
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 100000 > gpurun_out/r04b/phases_100k.json 2> gpurun_out/r04b/phases.err
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.timers.so timeout 300 python tools/fused_phases.py 8192 > gpurun_out/r04b/phases_8k.json 2>> gpurun_out/r04b/phases.err
PPSCI_HIP_LIB=/root/repo/paddlescience_amd/libppsci_hip.noslp.so timeout 300 python tools/fused_bench.py 100000 > gpurun_out/r04b/noslp.log 2>&1
cat gpurun_out/r04b/phases_100k.json gpurun_out/r04b/phases_8k.json gpurun_out/r04b/noslp.log; tail -3 gpurun_out/r04b/phases.err
