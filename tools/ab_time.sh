#!/bin/bash
# smr_kernel time (hipEvents, headline bench workload) of several library builds in one gpurun call (profiling builds
# give wrong results: only the kernel times are read).  usage: tools/ab_time.sh <out-prefix> <name> ...
pre=$1; shift
repo=$PWD
for round in 1 2; do
    for v in default "$@"; do
        lib=$repo/mrcaudiocodec_amd/libmrc_hip.so
        [ "$v" != default ] && lib=$repo/mrcaudiocodec_amd/libmrc_hip_$v.so
        MRC_HIP_LIBRARY=$lib timeout -k 10 120 python bench.py --frames 131072 --cpu-frames 0 --skip-extras 2>/dev/null | \
            python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $round, d['ms_per_step'], [(k['name'],k['ms']) for k in d['kernels']])" >> gpurun_out/${pre}_time.txt
    done
done
cat gpurun_out/${pre}_time.txt
