#!/bin/bash
# registers / LDS / occupancy of every kernel of one HIP source, as the compiler reports them (no GPU needed)
# usage: tools/kernel_resources.sh mrcaudiocodec_amd/csrc/mrc_kernels_long.hip [-D...]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$(dirname $0)/../include \
    "$@" \
    -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/_kr.o 2>&1 | python3 $(dirname $0)/kernel_resources.py
