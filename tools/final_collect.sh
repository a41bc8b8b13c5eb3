#!/bin/bash
# final profile pass of a round, on the GPU box (one gpurun call): kernel stats / traffic / SQ counters of the three single-GPU
# configurations, rocprofv3 --kernel-trace --stats of the chained call's two bench legs, node statistics, the default bench
# usage: tools/final_collect.sh <tag>          (outputs under gpurun_out/<tag>_*)
set -o pipefail
tag=${1:-r04f}
repo=$PWD; out=$repo/gpurun_out
bash tools/collect_counters.sh $tag 131072 "mono stereo switch" 1 > $out/${tag}_collect.log 2>&1; echo "collect rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_sm_stats -- python3 $repo/tools/stream_mode_only.py > $out/${tag}_sm.log 2>&1; echo "sm rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_ss_stats -- python3 $repo/tools/single_stream_bench.py --hops 65536 > $out/${tag}_ss.log 2>&1; echo "ss rc $?"
cd $repo
if [ -f mrcaudiocodec_amd/libmrc_hip_nodestats.so ]; then
    MRC_HIP_LIBRARY=$repo/mrcaudiocodec_amd/libmrc_hip_nodestats.so timeout -k 10 300 python tools/node_stats.py 2048 2>&1 | grep -v amdgpu.ids > $out/${tag}_node_stats.txt
fi
timeout -k 10 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc $?"
echo done
