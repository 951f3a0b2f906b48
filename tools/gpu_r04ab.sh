#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q 2>&1 | tail -5
