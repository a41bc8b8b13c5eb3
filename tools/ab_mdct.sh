#!/bin/bash
# tools/mdct_bench.py for several library builds in one gpurun call.  usage: tools/ab_mdct.sh <out-prefix> <name> ...
pre=$1; shift
repo=$PWD
for v in default "$@"; do
    lib=$repo/mrcaudiocodec_amd/libmrc_hip.so
    [ "$v" != default ] && lib=$repo/mrcaudiocodec_amd/libmrc_hip_$v.so
    echo "== $v" >> gpurun_out/${pre}_mdct.txt
    MRC_HIP_LIBRARY=$lib timeout -k 10 200 python tools/mdct_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-34s %8.4f ms  %6.1f GB/s  %.3f' % (d['case'], d['ms'], d['GBs'], d['frac_hbm']))" >> gpurun_out/${pre}_mdct.txt
done
cat gpurun_out/${pre}_mdct.txt
