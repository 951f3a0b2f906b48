#!/bin/bash
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r04s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04s/prof -o pn -- python /root/repo/tools/piratenet_step.py > /root/repo/gpurun_out/r04s/prof.log 2>&1
cd /root/repo
f=$(find gpurun_out/r04s/prof -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-140
