#!/bin/bash
# like tools/ab_time.sh, for the stereo (configs[2]) kernels: bench.py --only stereo per library build
pre=$1; shift
repo=$PWD
for round in 1 2; do
    for v in default "$@"; do
        lib=$repo/mrcaudiocodec_amd/libmrc_hip.so
        [ "$v" != default ] && lib=$repo/mrcaudiocodec_amd/libmrc_hip_$v.so
        MRC_HIP_LIBRARY=$lib timeout -k 10 120 python bench.py --frames 131072 --cpu-frames 0 --only stereo 2>/dev/null | tail -1 | sed "s/^/$v $round /" >> gpurun_out/${pre}_stereo.txt
    done
done
cat gpurun_out/${pre}_stereo.txt
