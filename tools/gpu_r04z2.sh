#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04z
timeout 1200 python bench.py > gpurun_out/r04z/bench.json 2> gpurun_out/r04z/bench.err
tail -c 200 gpurun_out/r04z/bench.err
