#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04p
for npx in 0 4 2 1 0; do
  PPSCI_PW_NPX=$npx timeout 300 python tools/tfno_step.py 50 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/npx=$npx /" | tee -a gpurun_out/r04p/tfno.log
done
