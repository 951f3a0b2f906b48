#!/bin/bash
# Run on the GPU box (via gpurun): dynamic instruction mix + stall counters of the Taylor kernels in bench.py's primary config.
# Counters in their own passes (no tracing domains together with --pmc); outputs under gpurun_out/insts_<tag>/
TAG=${1:-x}
OUT=/root/repo/gpurun_out/insts_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-strong"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o ac -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o ac -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p3 -o ac -- $CMD > $OUT/p3.log 2>&1
python /root/repo/tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
tail -5 $OUT/p1.log
cat $OUT/summary.txt
