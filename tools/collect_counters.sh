#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes over the bench command, outputs under gpurun_out/<tag>_*.
#   kernel-trace stats, FETCH_SIZE, WRITE_SIZE (separate --pmc passes, as MI355X_MICROARCH.md prescribes) and four SQ
#   counter sets for the issue / stall analysis.  Summaries are made afterwards by tools/summarize_profiles.py and
#   tools/summarize_sq.py from the merged gpurun_out/.
# usage: tools/collect_counters.sh <tag> <frames>
set -e -o pipefail
tag=$1; frames=${2:-65536}
repo=$PWD
out=$repo/gpurun_out
cmd="python3 $repo/bench.py --frames $frames --steps 3 --warmup 1 --cpu-frames 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- $cmd > $out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- $cmd > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- $cmd > $out/${tag}_write.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq$i -- $cmd > $out/${tag}_sq$i.log 2>&1 || echo "pass sq$i failed (see log)"
done
echo "collected $tag"
