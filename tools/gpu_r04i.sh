#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04i
timeout 2400 bash tools/profile_all.sh r04 > gpurun_out/r04i/profile_all.log 2>&1
ls gpurun_out/ | grep prof_r04
tail -5 gpurun_out/r04i/profile_all.log
