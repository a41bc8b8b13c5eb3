#!/bin/bash
# final profile pass, part 2: configs[3] again (short-block smr changed), the chained call's kernels
set -e -o pipefail
repo=$PWD; out=$repo/gpurun_out
bash tools/collect_counters.sh r03h 131072 "switch" 0 > $out/r03h_collect.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03h_sm_stats -- python3 $repo/tools/stream_mode_only.py > $out/r03h_sm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03h_ss_stats -- python3 $repo/tools/single_stream_bench.py --hops 65536 > $out/r03h_ss.log 2>&1
echo done
