#!/bin/bash
# All BASELINE configs under rocprofv3 (run through gpurun); summarise afterwards with tools/summarize_profile.py <tag> <stem>
R=${1:-r03}
bash /root/repo/tools/profile_bench.sh ${R}_bench
bash /root/repo/tools/profile_bench.sh ${R}_ns python /root/repo/tools/ns_step.py 20
bash /root/repo/tools/profile_bench.sh ${R}_laplace python /root/repo/tools/laplace_step.py 200
bash /root/repo/tools/profile_bench.sh ${R}_tfno python /root/repo/tools/tfno_step.py 30
bash /root/repo/tools/profile_bench.sh ${R}_uno python /root/repo/tools/uno_step.py 30
bash /root/repo/tools/profile_bench.sh ${R}_sfno python /root/repo/tools/sfno_step.py 30
bash /root/repo/tools/profile_bench.sh ${R}_spinn python /root/repo/tools/spinn_step.py 50
bash /root/repo/tools/profile_bench.sh ${R}_piratenet python /root/repo/tools/piratenet_step.py
bash /root/repo/tools/profile_bench.sh ${R}_ac256 python /root/repo/tools/ac256_step.py 10
