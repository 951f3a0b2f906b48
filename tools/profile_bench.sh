#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes (each in its own run: no tracing domains together with
# --pmc) of one command; outputs under gpurun_out/prof_<tag>/ for tools/summarize_profile.py.
#   profile_bench.sh <tag> [command ...]      default command: the primary bench config
TAG=${1:-r03}
shift
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $# -gt 0 ]; then CMD="$*"; else CMD="python /root/repo/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-strong"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ac -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o ac -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o ac -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma -o ac -- $CMD > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_wait -o ac -- $CMD > $OUT/pmc_wait.log 2>&1
find $OUT -name "*_kernel_stats.csv" | head -3
