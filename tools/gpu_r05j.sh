#!/bin/bash
# round 5, final build: rocprofv3 profiles of every config (kernel trace + PMC passes, each in its own run)
cd /root/repo
O=gpurun_out/r05j
mkdir -p $O
bash tools/profile_all.sh r05 > $O/profile.log 2>&1
tail -3 $O/profile.log
ls gpurun_out | grep prof_r05
