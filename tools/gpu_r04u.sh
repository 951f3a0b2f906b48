#!/bin/bash
bash /root/repo/tools/profile_bench.sh r04_tfno python /root/repo/tools/tfno_step.py 30
bash /root/repo/tools/profile_bench.sh r04_piratenet python /root/repo/tools/piratenet_step.py
