#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04z
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04z/tests.log 2>&1
tail -3 gpurun_out/r04z/tests.log
timeout 1200 python bench.py > gpurun_out/r04z/bench.json 2> gpurun_out/r04z/bench.err
tail -c 300 gpurun_out/r04z/bench.err
bash /root/repo/tools/profile_bench.sh r04_tfno python /root/repo/tools/tfno_step.py 30 > /dev/null 2>&1
