#!/bin/bash
# SQ counters of the per-shape kernels (tools/smr_shapes_bench.py under rocprofv3 --pmc): is a shape's smr_kernel / MDCT bound
# by VALU issue or by latency?  outputs under gpurun_out/<tag>_shapes_sq*
set -e -o pipefail
tag=$1; repo=$PWD; out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/${tag}_shapes_sq1 -- python3 $repo/tools/smr_shapes_bench.py > $out/${tag}_shapes_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/${tag}_shapes_sq2 -- python3 $repo/tools/smr_shapes_bench.py > $out/${tag}_shapes_sq2.log 2>&1
echo done
