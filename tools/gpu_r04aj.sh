#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_golden_piratenet.py tests/test_golden_modified_mlp.py tests/test_examples.py tests/test_layerwise_mlp.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python tools/piratenet_step.py 2>&1 | grep -v amdgpu.ids | tail -1; done
