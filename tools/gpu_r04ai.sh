#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_fno.py tests/test_fno_native.py tests/test_fno_net.py tests/test_golden_fno.py tests/test_fullsize.py tests/test_neuralop_data.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python tools/tfno_step.py 100 2>&1 | grep -v amdgpu.ids | tail -1; done
