#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes over the bench command, outputs under gpurun_out/<tag>_*.
#   kernel-trace stats, FETCH_SIZE, WRITE_SIZE (separate --pmc passes, as MI355X_MICROARCH.md prescribes) for the three
#   single-GPU configurations (mono = configs[1], the bench headline; stereo = configs[2]; switch = configs[3]) and four
#   SQ counter sets for the issue / stall analysis of the mono run.  Summaries are made afterwards by
#   tools/summarize_profiles.py and tools/summarize_sq.py from the merged gpurun_out/.
# usage: tools/collect_counters.sh <tag> <frames> [configs: "mono stereo switch"] [sq: 1|0]
set -e -o pipefail
tag=$1; frames=${2:-65536}; configs=${3:-"mono stereo switch"}; sq=${4:-1}
repo=$PWD
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in $configs; do
    case $cfg in
        mono)   extra="--skip-extras" ;;
        stereo) extra="--only stereo" ;;
        switch) extra="--only switch" ;;
    esac
    cmd="python3 $repo/bench.py --frames $frames --steps 3 --warmup 1 --cpu-frames 0 $extra"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${cfg}_stats -- $cmd > $out/${tag}_${cfg}_stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_${cfg}_fetch -- $cmd > $out/${tag}_${cfg}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_${cfg}_write -- $cmd > $out/${tag}_${cfg}_write.log 2>&1
    echo "collected $tag $cfg"
done
if [ "$sq" = "1" ]; then
    cmd="python3 $repo/bench.py --frames $frames --steps 3 --warmup 1 --cpu-frames 0 --skip-extras"
    i=0
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
               "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
               "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
               "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
        i=$((i+1))
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq$i -- $cmd > $out/${tag}_sq$i.log 2>&1 || echo "pass sq$i failed (see log)"
    done
fi
echo "collected $tag"
