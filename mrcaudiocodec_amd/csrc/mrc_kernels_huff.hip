// Huffman table PRICING on the device (codecThem.py:136-180): which of the four trained tables -- or raw
// mantissas, id 15 -- costs the fewest bits for one channel chunk, and the reservoir credit
// bits_saved = raw - best (codecThem.py:202).  Only the bit COUNTS are needed to carry the bit reservoir to the
// next block (codecThem.py:224,274), so a chained multi-stream encode can stay on the GPU from block to block;
// the code strings / packed bytes are produced afterwards on the host (csrc/mrc_pack.cpp, which prices the
// same way -- tests compare the two).
//
// One wavefront per (frame, stream); lanes stride over the lines.  The reference's pricing quirk is kept: a
// mantissa equal to the table's escape VALUE is priced as its code alone (codecThem.py:169-172).
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
namespace {

constexpr int kLutSize = 65;                           // largest value in any table is 64
// code length per value (0 = not in the table); rows: percussive, silence, speech, tonal (sorted names)
__constant__ unsigned char kCodeLen[4][kLutSize] = {
    {1, 4, 3, 6, 3, 4, 6, 8, 5, 6, 7, 7, 7, 9, 9, 0, 6},
    {2, 3, 3, 5, 2, 4, 6, 0, 5, 5, 6, 4},
    {2, 4, 3, 6, 2, 4, 6, 4, 4, 5, 6, 7, 7, 0, 0, 0, 6, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7},
    {1, 5, 3, 7, 3, 7, 8, 4, 4, 7, 8, 0, 0, 0, 0, 0, 5, 7, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6,
     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8}};
__constant__ int kEscape[4] = {16, 11, 7, 7};

// sum over the 64 lanes, in registers (DPP inside a 16-lane row, gfx950 permlane swaps across rows); all lanes active
__device__ __forceinline__ int wave_sum_int(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);     // row_ror:8
    const uint2v r16 = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(r16.x + r16.y);
    const uint2v r32 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r32.x + r32.y);
}

// grid: nFrames blocks of nStreams waves.  reservoirNext (may be null): reservoirOut[f] + sum over the frame's
// streams of bits_saved -- the value the NEXT block of the same stream starts from (codecThem.py:274).
__global__ void huffman_gain_kernel(DevShape S, int nStreams, const int* __restrict__ bitAlloc,
                                    const int* __restrict__ mantissa, const int* __restrict__ reservoirOut,
                                    int* __restrict__ huffTable, int* __restrict__ bitsSaved,
                                    int* __restrict__ reservoirNext) {
    __shared__ int sSaved[4];
    const int lane = threadIdx.x & (kWave - 1), strm = threadIdx.x >> 6;
    const int64_t f = blockIdx.x;
    const int M = S.halfN, nb = S.nBands;
    const int* ba = bitAlloc + (f * nStreams + strm) * nb;
    const int* m = mantissa + (f * nStreams + strm) * (int64_t)M;
    int raw = 0, cost[4] = {0, 0, 0, 0};
    for (int k = lane; k < M; k += kWave) {
        const int b = ba[S.bandOfLine[k]];
        if (b) {
            const int v = m[k];
            raw += b;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int len = (v >= 0 && v < kLutSize) ? kCodeLen[t][v] : 0;
                cost[t] += len ? len : b + kCodeLen[t][kEscape[t]];
            }
        }
    }
    raw = wave_sum_int(raw);
    int best = raw, table = 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = wave_sum_int(cost[t]);
        if (c < best) { best = c; table = t; }               // strictly less: the first table wins ties, raw wins ties
    }
    if (lane == 0) {
        huffTable[f * nStreams + strm] = table;
        bitsSaved[f * nStreams + strm] = raw - best;
        sSaved[strm] = raw - best;
    }
    __syncthreads();
    if (reservoirNext && threadIdx.x == 0) {
        int r = reservoirOut[f];
        for (int s = 0; s < nStreams; ++s) r += sSaved[s];
        reservoirNext[f] = r;
    }
}

}  // namespace

hipError_t launch_huffman_gain(const DevShape& S, int64_t nFrames, int nStreams, const int* bitAlloc,
                               const int* mantissa, const int* reservoirOut, int* huffTable, int* bitsSaved,
                               int* reservoirNext, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    hipLaunchKernelGGL(huffman_gain_kernel, dim3((unsigned)nFrames), dim3(kWave * nStreams), 0, st, S, nStreams,
                       bitAlloc, mantissa, reservoirOut, huffTable, bitsSaved, reservoirNext);
    return hipGetLastError();
}

}  // namespace mrc
