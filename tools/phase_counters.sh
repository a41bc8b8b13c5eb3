#!/bin/bash
# Runs on the GPU box: one SQ counter pass of the headline bench per profiling build of the library
# (libmrc_hip_<name>.so: smr_kernel cut after phase n, without the sweep, ...): where the kernel's VALU and LDS
# instructions are.  usage: tools/phase_counters.sh <tag> <frames> <name> ...   -> gpurun_out/<tag>_<name>_pc/
set -o pipefail
tag=$1; frames=$2; shift 2
repo=$PWD
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
    lib=$repo/mrcaudiocodec_amd/libmrc_hip.so
    [ "$v" != default ] && lib=$repo/mrcaudiocodec_amd/libmrc_hip_$v.so
    export MRC_HIP_LIBRARY=$lib
    timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES \
        --kernel-trace --output-format csv -d $out/${tag}_${v}_pc -- python3 $repo/bench.py --frames $frames --steps 2 --warmup 1 --cpu-frames 0 --skip-extras > $out/${tag}_${v}_pc.log 2>&1 || echo "$v failed"
    echo "done $v"
done
