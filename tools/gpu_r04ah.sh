#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_fused_step.py tests/test_one_launch.py tests/test_fullsize.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python tools/fused_bench.py 100000 2>&1 | grep net; done
