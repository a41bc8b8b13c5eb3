#!/bin/bash
# GPU box: SQ_INSTS_VALU etc. of the bench command for each library variant given (profiling builds of
# mrcaudiocodec_amd/csrc with EXTRA=-DMRC_PROFILE_SKIP=<mask>), outputs under gpurun_out/var_<name>.
# usage: tools/collect_variants.sh <frames> <variant.so> ...
set -e -o pipefail
frames=$1; shift
repo=$PWD; out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
    name=$(basename $lib .so)
    MRC_HIP_LIBRARY=$repo/$lib rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
        --kernel-trace --output-format csv -d $out/var_$name -- python3 $repo/bench.py --frames $frames --steps 2 --warmup 1 --cpu-frames 0 \
        > $out/var_$name.log 2>&1
done
echo "variants done"
