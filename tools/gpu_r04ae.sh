#!/bin/bash
cd /root/repo
for i in 1 2 3; do timeout 300 python tools/tfno_step.py 100 2>&1 | grep -v amdgpu.ids | tail -1; done
