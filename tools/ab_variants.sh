#!/bin/bash
# A/B of library build variants on the GPU box, all in one gpurun call: for each libmrc_hip_<name>.so (and the default
# library) the headline bench (resident mono, --skip-extras) three times interleaved, then a short parity sweep.
# usage: tools/ab_variants.sh <out-prefix> <name> [<name> ...]
set -o pipefail
pre=$1; shift
repo=$PWD
for round in 1 2 3; do
    for v in default "$@"; do
        lib=$repo/mrcaudiocodec_amd/libmrc_hip.so
        [ "$v" != default ] && lib=$repo/mrcaudiocodec_amd/libmrc_hip_$v.so
        MRC_HIP_LIBRARY=$lib timeout -k 10 120 python bench.py --frames 131072 --cpu-frames 0 --skip-extras 2>/dev/null | \
            python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $round, d['value'], d['ms_per_step'], [(k['name'],k['ms']) for k in d['kernels']])" >> gpurun_out/${pre}_ab.txt
    done
done
for v in "$@"; do
    MRC_HIP_LIBRARY=$repo/mrcaudiocodec_amd/libmrc_hip_$v.so timeout -k 10 200 python tools/parity_sweep.py --mono 4096 --joint 1024 --varied 2048 2>&1 | grep RESULT | sed "s/^/$v /" >> gpurun_out/${pre}_ab.txt
done
cat gpurun_out/${pre}_ab.txt
