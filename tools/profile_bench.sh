#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py; outputs under gpurun_out/prof_<tag>/
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-strong"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ac -- $CMD > $OUT/trace.log 2>&1
# counters in their own runs (no tracing domains together with --pmc)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o ac -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o ac -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -o ac -- $CMD > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/pmc_wait -o ac -- $CMD > $OUT/pmc_wait.log 2>&1
find $OUT -name "*.csv" | head -30
