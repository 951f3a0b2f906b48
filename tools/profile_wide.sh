#!/bin/bash
# Run on the GPU box (via gpurun): instruction mix / stall / LDS counters of the feature-split kernels on the NS 5x128 shard
# (tools/wide_bench.py).  Counters in their own passes; outputs under gpurun_out/wide_<tag>/
TAG=${1:-x}
OUT=/root/repo/gpurun_out/wide_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/tools/wide_bench.py --ns-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ns -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o ns -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o ns -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM --output-format csv -d $OUT/p3 -o ns -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -o ns -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -o ns -- $CMD > $OUT/p5.log 2>&1
python /root/repo/tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -12
