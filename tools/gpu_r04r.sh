#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04r
timeout 300 python tools/piratenet_step.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r04r/pirate.log
timeout 300 python tools/tfno_step.py 50 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r04r/tfno.log
timeout 300 python tools/piratenet_step.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r04r/pirate.log
