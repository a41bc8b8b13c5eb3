// `.pac` chunk packing ON THE DEVICE: the bytes WriteDataBlock / JointWriteDataBlock append per block
// (pacfileThem.py:652-781, 825-963; bit order of bitpack.py:36-101; Huffman choice and recoding of
// codecThem.py:136-203), produced from the encoder's outputs where they lie in HBM -- so that a WAV -> .pac
// pipeline copies ~350 bytes per block back to the host instead of the 2 KB mantissa plane, and no host thread
// touches a bit.  Same byte image as the host packer (csrc/mrc_pack.cpp), which the tests compare it with.
//
// Three steps, all wave-granular (one 64-lane wavefront per channel chunk, no workgroup barrier after the
// table staging):
//   pack_plan_kernel    per chunk: price the four Huffman tables and raw in ONE pass over the mantissas (or take
//                       the table from huffman_gain_kernel / the caller), count the bits the writer will emit
//                       -> table id, bits_saved, chunk size in bytes (pacfileThem.py:706-707);
//   pack_scan_*         exclusive prefix sum of (4 + chunk bytes) -> where every chunk starts, block offsets;
//   pack_write_kernel   per chunk: per-line code lengths -> prefix sum over the lines in LDS -> every lane ORs
//                       its codes into a zeroed big-endian word image of the chunk in LDS (ds_or_b32; a code of
//                       at most 25 bits touches at most two words) -> band headers at the positions the prefix
//                       sum gives -> the image goes out as bytes behind the 4-byte little-endian length.
// Line k's first bit sits at  header bits + (band(k) + 1) (nMantSizeBits + nScaleBits) + sum of the code
// lengths of the lines before it: the band headers are a closed-form term, so ONE prefix sum serves lines and
// headers alike.
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
const int* pack_error_flag(const void* ws, int64_t nChunks);
const long long* pack_total_bytes(const void* ws, int64_t nChunks);
namespace {

constexpr int kWavesPerGroup = 4;
constexpr int kScanTile = 1024;                      // chunks per workgroup of the position scan

__device__ __forceinline__ int wave_sum_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);     // row_ror:8
    const uint2v r16 = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(r16.x + r16.y);
    const uint2v r32 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r32.x + r32.y);
}
// inclusive prefix sum over the 64 lanes (Kogge-Stone in 16-lane rows by DPP shifts, row totals by row_bcast)
__device__ __forceinline__ int wave_scan_i(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, true);            // row_shr:1 (bound_ctrl: 0 shifted in)
    v += __builtin_amdgcn_mov_dpp(v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
// LDS traffic between the lanes of ONE wave: order this wave's DS operations (no other wave is involved)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ unsigned lut_index(int v) { return (unsigned)v < (unsigned)kPackLutSize ? (unsigned)v : (unsigned)kPackLutSize; }

// what the writer emits for one mantissa under `table` (pacfileThem.py:685-703, 742-760): -> (bits, length)
__device__ __forceinline__ void code_of(const unsigned* __restrict__ emit, int table, int v, int ba, unsigned* bits, int* len) {
    if (table == kPackRawTable) { *bits = (unsigned)v & ((1u << ba) - 1u); *len = ba; return; }
    const unsigned e = emit[table * (kPackLutSize + 1) + lut_index(v)];
    const int n = (int)((e >> 16) & 0x7fffu);
    if (e >> 31) { *bits = ((e & 0xffffu) << ba) | ((unsigned)v & ((1u << ba) - 1u)); *len = n + ba; }   // escape code + raw mantissa
    else { *bits = e & 0xffffu; *len = n; }
}

template <class MantT>
__global__ __launch_bounds__(kWave * kWavesPerGroup) void pack_plan_kernel(
    DevShape S, PackParams P, PackTables T, int64_t nChunks, const int* __restrict__ bitAlloc,
    const MantT* __restrict__ mant, const int* __restrict__ tableIn, int* __restrict__ tableOut,
    int* __restrict__ bitsSaved, int* __restrict__ chunkBytes, int* __restrict__ errorFlag, int fast16,
    const long long* __restrict__ chunkMap /* nullable: this group's chunk -> global chunk index */) {
    __shared__ unsigned sEmit[4 * (kPackLutSize + 1)];
    __shared__ int sBa[kWavesPerGroup * kMaxBands];
    for (int i = threadIdx.x; i < 4 * (kPackLutSize + 1); i += blockDim.x) sEmit[i] = T.emit[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + wave;
    if (c >= nChunks) return;
    const int M = S.halfN, nb = S.nBands;
    const int* ba = bitAlloc + c * nb;
    const MantT* m = mant + c * (int64_t)M;
    // ONE pass: the raw size, the PRICE of each table (codecThem.py:161-180: a value without a code costs the escape
    // code + the raw mantissa, the escape VALUE itself its code alone) and what the WRITER emits for it (194-200: the
    // escape value is followed by its raw mantissa too)
    int raw = 0, cost[4] = {0, 0, 0, 0}, wr[4] = {0, 0, 0, 0};
    auto price = [&](int v, int b) {
        const unsigned idx = lut_index(v);
        raw += b;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned e = sEmit[t * (kPackLutSize + 1) + idx];
            const int n = (int)((e >> 16) & 0x7fffu);
            const bool follows = (e >> 31) != 0;
            wr[t] += n + (follows ? b : 0);
            cost[t] += n + ((follows && (int)idx != T.escape[t]) ? b : 0);
        }
    };
    if (fast16) {
        // 1024-line chunks: 16 consecutive lines per lane -- one 16-byte load of their bands, two / four of their
        // mantissas, the bit allocation staged in LDS (see pack_write_kernel)
        int* baL = sBa + wave * kMaxBands;
        if (lane < nb) baL[lane] = ba[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int k0 = 16 * lane;
        const uint4 bw = *reinterpret_cast<const uint4*>(S.bandOfLine + k0);
        const unsigned bwv[4] = {bw.x, bw.y, bw.z, bw.w};
        int v[16];
        if (sizeof(MantT) == 2) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(m + k0), a1 = *reinterpret_cast<const uint4*>(m + k0 + 8);
            const unsigned aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (int)((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 a = *reinterpret_cast<const int4*>(m + k0 + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = baL[(bwv[j >> 2] >> (8 * (j & 3))) & 0xffu];
            if (b) price(v[j], b);
        }
    } else {
        for (int k = lane; k < M; k += kWave) {
            const int b = ba[S.bandOfLine[k]];
            if (b) price((int)m[k], b);
        }
    }
    raw = wave_sum_i(raw);
    int best = raw, table = kPackRawTable, mantBits = raw, saved = 0;
    int wsum[4], csum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { csum[t] = wave_sum_i(cost[t]); wsum[t] = wave_sum_i(wr[t]); }
    if (tableIn) {                                                   // chosen elsewhere (huffman_gain_kernel / the caller)
        table = tableIn[c];
        if (table != kPackRawTable && (table < 0 || table > 3)) { table = kPackRawTable; if (lane == 0) atomicOr(errorFlag, 1); }
    } else if (P.useHuffman) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (csum[t] < best) { best = csum[t]; table = t; }         // strictly less: raw, then the first table, win ties
        saved = raw - best;                                          // codecThem.py:202
    }
    if (table != kPackRawTable) mantBits = table == 0 ? wsum[0] : table == 1 ? wsum[1] : table == 2 ? wsum[2] : wsum[3];
    const int ch = (int)(c % P.nch);
    int bits = 4 + P.blkBitsA + P.blkBitsB + nb * (P.nMantSizeBits + P.nScaleBits) + mantBits;
    if (P.joint) { if (ch == 0) bits += nb + 4 * P.nScaleBits; }      // pacfileThem.py:826-833
    else bits += P.nScaleBits;                                       // pacfileThem.py:655
    if (lane == 0) {
        chunkBytes[chunkMap ? chunkMap[c] : c] = (bits + 7) / 8;     // pacfileThem.py:706-707
        tableOut[c] = table;
        if (bitsSaved) bitsSaved[c] = saved;
    }
}

// ---- positions: pos[c] = sum over c' < c of (4 + chunkBytes[c']) ------------------------------------------------
__global__ __launch_bounds__(256) void pack_scan_sums_kernel(int64_t n, const int* __restrict__ chunkBytes,
                                                              long long* __restrict__ tileSum) {
    __shared__ int sWave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + 4 * threadIdx.x;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) s += 4 + chunkBytes[base + j];
    s = wave_sum_i(s);
    if ((threadIdx.x & 63) == 0) sWave[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tileSum[blockIdx.x] = (long long)sWave[0] + sWave[1] + sWave[2] + sWave[3];
}
__global__ void pack_scan_tiles_kernel(int nTiles, long long* __restrict__ tileSum, long long* __restrict__ total) {
    if (threadIdx.x || blockIdx.x) return;                           // a few thousand tiles at most: one lane walks them
    long long run = 0;
    for (int g = 0; g < nTiles; ++g) { const long long t = tileSum[g]; tileSum[g] = run; run += t; }
    *total = run;
}
__global__ __launch_bounds__(256) void pack_scan_apply_kernel(int64_t n, int nch, const int* __restrict__ chunkBytes,
                                                               const long long* __restrict__ tileSum,
                                                               const long long* __restrict__ total,
                                                               long long* __restrict__ pos, long long* __restrict__ blockOffset,
                                                               const int* __restrict__ chunkStream, int hdrLen) {
    __shared__ int sWave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + 4 * threadIdx.x;
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = base + j < n ? 4 + chunkBytes[base + j] : 0; s += v[j]; }
    const int incl = wave_scan_i(s);
    if ((threadIdx.x & 63) == 63) sWave[threadIdx.x >> 6] = incl;
    __syncthreads();
    long long run = tileSum[blockIdx.x] + (incl - s);
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += sWave[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = base + j;
        if (c < n) {
            // chunkStream: a file header of hdrLen bytes precedes the first chunk of every stream
            const long long at = run + (chunkStream ? (long long)hdrLen * (chunkStream[c] + 1) : 0);
            pos[c] = at;
            if (nch > 0 && c % nch == 0) blockOffset[c / nch] = at;
        }
        run += v[j];
    }
    if (nch > 0 && blockIdx.x == 0 && threadIdx.x == 0) blockOffset[n / nch] = *total;
}
// (after pack_scan_apply_kernel has read it) the total with the headers in, and the end position behind the last chunk
__global__ void pack_scan_finish_kernel(int64_t n, long long* __restrict__ total, long long* __restrict__ pos,
                                        const int* __restrict__ chunkStream, int hdrLen) {
    if (threadIdx.x || blockIdx.x) return;
    if (chunkStream && n > 0) *total += (long long)hdrLen * (chunkStream[n - 1] + 1);
    pos[n] = *total;
}

// ---- payloads ---------------------------------------------------------------------------------------------------
// OR the lowest `len` (1..25) bits of `val` into the MSB-first bit stream held as big-endian 32-bit words
__device__ __forceinline__ void put_bits(unsigned* __restrict__ w, int off, unsigned val, int len) {
    const int wi = off >> 5, sh = 32 - (off & 31) - len;
    if (sh >= 0) atomicOr(&w[wi], val << sh);
    else { atomicOr(&w[wi], val >> (-sh)); atomicOr(&w[wi + 1], val << (32 + sh)); }
}
__device__ __forceinline__ int pad16(int k) { return k + (k >> 4); }     // (a lane's run of 16 entries starts on its own bank)

template <class MantT>
__global__ __launch_bounds__(kWave * kWavesPerGroup) void pack_write_kernel(
    DevShape S, PackParams P, PackTables T, int64_t nChunks, const int* __restrict__ oscale,
    const int* __restrict__ msSwitch, const int* __restrict__ scaleFactor, const int* __restrict__ bitAlloc,
    const MantT* __restrict__ mant, const int* __restrict__ table, const int* __restrict__ chunkBytes,
    const long long* __restrict__ pos, unsigned char* __restrict__ out, long long outCap, int wordsPerWave,
    int fast16 /* 1024 lines, every band non-empty, planes 16-byte aligned */,
    const long long* __restrict__ chunkMap, int* __restrict__ errorFlag) {
    extern __shared__ unsigned smem[];
    unsigned* sEmit = smem;                                                     // [4 (kPackLutSize + 1)]
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int M = S.halfN, nb = S.nBands;
    const int prefLen = pad16(M) + 2;
    unsigned* w = smem + 4 * (kPackLutSize + 1) + wave * (wordsPerWave + prefLen + kWave);   // the chunk's image
    int* pref = reinterpret_cast<int*>(w + wordsPerWave);                       // [pad16(M) + 1] code lengths -> prefix sums
    int* laneBase = pref + prefLen;                                             // [64]
    for (int i = threadIdx.x; i < 4 * (kPackLutSize + 1); i += blockDim.x) sEmit[i] = T.emit[i];
    for (int i = lane; i < wordsPerWave; i += kWave) w[i] = 0u;
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * kWavesPerGroup + wave;
    if (c >= nChunks) return;
    const int64_t blk = c / P.nch;
    const int ch = (int)(c % P.nch);
    const int64_t gc = chunkMap ? chunkMap[c] : c;
    const int nBytes = chunkBytes[gc];
    const long long p0 = pos[gc];
    if (p0 + 4 + nBytes > outCap || nBytes > 4 * (wordsPerWave - 1)) {          // nothing is written: the caller learns why
        // 2: a chunk larger than mrc_pack_bound allows (bit_alloc > 16 handed in), 4: out_cap exceeded
        if (lane == 0) atomicOr(errorFlag, nBytes > 4 * (wordsPerWave - 1) ? 2 : 4);
        return;
    }
    const int tbl = table[c];
    const int* ba = bitAlloc + c * nb;
    const int* sf = scaleFactor + c * nb;
    const MantT* m = mant + c * (int64_t)M;
    const int hb = P.nMantSizeBits + P.nScaleBits;
    // chunk header: table id, block-switching bits, overall scale(s), M/S switch bits (pacfileThem.py:716-728, 892-910)
    int hdr = 4 + P.blkBitsA + P.blkBitsB;
    if (lane == 0) {
        put_bits(w, 0, (unsigned)tbl & 15u, 4);
        if (P.blkBitsA) put_bits(w, 4, P.bitA & ((1u << P.blkBitsA) - 1u), P.blkBitsA);
        if (P.blkBitsB) put_bits(w, 4 + P.blkBitsA, P.bitB & ((1u << P.blkBitsB) - 1u), P.blkBitsB);
    }
    const unsigned scaleMask = (1u << P.nScaleBits) - 1u;
    if (P.joint) {
        if (ch == 0) {
            if (lane < 4) put_bits(w, hdr + lane * P.nScaleBits, (unsigned)oscale[blk * 4 + lane] & scaleMask, P.nScaleBits);   // L, R, M, S
            if (lane < nb) put_bits(w, hdr + 4 * P.nScaleBits + lane, (unsigned)msSwitch[blk * nb + lane] & 1u, 1);
            hdr += 4 * P.nScaleBits + nb;
        }
    } else {
        if (lane == 0) put_bits(w, hdr, (unsigned)oscale[blk * P.nch + ch] & scaleMask, P.nScaleBits);
        hdr += P.nScaleBits;
    }
    // FAST PATH, 1024-line chunks (the long blocks: nearly all of a stream): a lane owns 16 CONSECUTIVE lines from the
    // start -- their bands are one 16-byte load, their mantissas two (uint16) or four (int32), codes, lengths and the
    // prefix over the 16 stay in registers, only the 64 lane totals are scanned across the wave and only the band starts
    // go through LDS (for the band headers).  No second pass over the mantissas, no length array in LDS.
    if (fast16) {
        int* baL = laneBase;                                                    // [32] bit allocation per band
        int* bandPref = laneBase + 32;                                          // [32] bits of the lines before the band's first
        if (lane < nb) baL[lane] = ba[lane];
        wave_sync();
        const int k0 = 16 * lane;
        const uint4 bw = *reinterpret_cast<const uint4*>(S.bandOfLine + k0);
        const unsigned bwv[4] = {bw.x, bw.y, bw.z, bw.w};
        const int bandBefore = lane ? (int)S.bandOfLine[k0 - 1] : -1;
        int v[16];
        if (sizeof(MantT) == 2) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(m + k0), a1 = *reinterpret_cast<const uint4*>(m + k0 + 8);
            const unsigned aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (int)((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 a = *reinterpret_cast<const int4*>(m + k0 + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
        }
        unsigned bits[16];
        int len[16], bandOf[16], run = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            bandOf[j] = (int)((bwv[j >> 2] >> (8 * (j & 3))) & 0xffu);
            const int b = baL[bandOf[j]];
            bits[j] = 0u; len[j] = 0;
            if (b) code_of(sEmit, tbl, v[j], b, &bits[j], &len[j]);
            run += len[j];
        }
        const int base = wave_scan_i(run) - run;
        int at = base;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (bandOf[j] != (j ? bandOf[j - 1] : bandBefore)) bandPref[bandOf[j]] = at;   // this line starts its band
            if (len[j]) put_bits(w, hdr + (bandOf[j] + 1) * hb + at, bits[j], len[j]);
            at += len[j];
        }
        wave_sync();
        if (lane < nb) {
            const int b = baL[lane];
            const unsigned hv = ((unsigned)(b ? b - 1 : 0) << P.nScaleBits) | ((unsigned)sf[lane] & scaleMask);
            put_bits(w, hdr + lane * hb + bandPref[lane], hv & ((1u << hb) - 1u), hb);
        }
        wave_sync();
        unsigned char* dstF = out + p0;
        if (lane < 4) dstF[lane] = (unsigned char)(((unsigned)nBytes >> (8 * lane)) & 255u);
        for (int j = lane; j < nBytes; j += kWave) dstF[4 + j] = (unsigned char)((w[j >> 2] >> (24 - 8 * (j & 3))) & 255u);
        return;
    }
    // code length of every line -> LDS (coalesced over the lines), then an exclusive prefix sum: lane l scans the run
    // [l LPL, (l + 1) LPL) in place, the lane totals are scanned in registers
    for (int k = lane; k < M; k += kWave) {
        const int b = ba[S.bandOfLine[k]];
        int len = 0;
        if (b) { unsigned bits; code_of(sEmit, tbl, (int)m[k], b, &bits, &len); }
        pref[pad16(k)] = len;
    }
    wave_sync();
    const int lpl = (M + kWave - 1) / kWave;
    {
        int run = 0;
        const int k0 = lane * lpl, k1 = min(k0 + lpl, M);
        for (int k = k0; k < k1; ++k) { const int t = pref[pad16(k)]; pref[pad16(k)] = run; run += t; }
        const int incl = wave_scan_i(run);
        laneBase[lane] = incl - run;
        if (lane == kWave - 1) pref[pad16(M) + 1] = incl;            // the total: the prefix of a band that starts at M
    }
    wave_sync();
    auto prefix_at = [&](int k) { return k < M ? pref[pad16(k)] + laneBase[k / lpl] : pref[pad16(M) + 1]; };
    // mantissas.  A lane takes a RUN of consecutive lines (the one it scanned): the 64 lanes of one ds_or then hit 64
    // different words -- with k = lane + 64 j about eight neighbouring lines share a word and the atomics serialise
    for (int k = lane * lpl, kEnd = min(k + lpl, M); k < kEnd; ++k) {
        const int band = S.bandOfLine[k];
        const int b = ba[band];
        if (b) {
            unsigned bits; int len;
            code_of(sEmit, tbl, (int)m[k], b, &bits, &len);
            if (len) put_bits(w, hdr + (band + 1) * hb + prefix_at(k), bits, len);
        }
    }
    // band headers: bit allocation (stored one lower) and scale factor (pacfileThem.py:730-732)
    if (lane < nb) {
        const int b = ba[lane];
        const unsigned v = ((unsigned)(b ? b - 1 : 0) << P.nScaleBits) | ((unsigned)sf[lane] & scaleMask);
        put_bits(w, hdr + lane * hb + prefix_at(S.bandLo[lane]), v & ((1u << hb) - 1u), hb);
    }
    wave_sync();
    // out: 4-byte little-endian length, then the image MSB first
    unsigned char* dst = out + p0;
    if (lane < 4) dst[lane] = (unsigned char)(((unsigned)nBytes >> (8 * lane)) & 255u);
    for (int j = lane; j < nBytes; j += kWave) dst[4 + j] = (unsigned char)((w[j >> 2] >> (24 - 8 * (j & 3))) & 255u);
}

// total and error flag straight into page-locked host memory (device-visible): no copy command on the kernel stream
__global__ void pack_export_kernel(const long long* __restrict__ total, const int* __restrict__ errorFlag,
                                   long long* __restrict__ hostOut) {
    if (threadIdx.x || blockIdx.x) return;
    hostOut[0] = *total;
    hostOut[1] = *errorFlag;
    __threadfence_system();
}

}  // namespace

hipError_t launch_pack_export(const void* ws, int64_t nChunks, long long* hostOut, hipStream_t st) {
    hipLaunchKernelGGL(pack_export_kernel, dim3(1), dim3(64), 0, st, pack_total_bytes(ws, nChunks), pack_error_flag(ws, nChunks), hostOut);
    return hipGetLastError();
}

size_t pack_workspace_bytes(int64_t nChunks) {
    const int64_t nTiles = (nChunks + kScanTile - 1) / kScanTile;
    return (size_t)(nChunks + 1) * (sizeof(int) + sizeof(long long)) + (size_t)(nTiles + 2) * sizeof(long long) + 64;
}

// workspace: pos [nChunks + 1] | tile sums [nTiles] | total | error flag | chunkBytes [nChunks]
PackWs pack_ws_views(void* ws, int64_t nChunks) {
    const int64_t nTiles = (nChunks + kScanTile - 1) / kScanTile;
    PackWs W;
    W.pos = reinterpret_cast<long long*>(ws);
    W.tileSum = W.pos + nChunks + 1;
    W.total = W.tileSum + nTiles;
    W.errorFlag = reinterpret_cast<int*>(W.total + 1);
    W.chunkBytes = W.errorFlag + 2;
    return W;
}

static int pack_fast16(const DevShape& S, const void* mant, size_t mantSize, bool packFast16Ok) {
    return S.halfN == 16 * kWave && S.nBands <= 32 && packFast16Ok && !(reinterpret_cast<uintptr_t>(mant) & 15) &&
           ((size_t)S.halfN * mantSize) % 16 == 0;
}

hipError_t launch_pack_plan(const DevShape& S, const PackParams& P, const PackTables& T, int64_t nBlocks,
                            const int* bitAlloc, const void* mant, int mantFmt, const int* tableIn, int* tableOut,
                            int* bitsSaved, const PackWs& W, const long long* chunkMap, bool allBandsNonEmpty,
                            hipStream_t st) {
    const int64_t nChunks = nBlocks * P.nch;
    if (nChunks <= 0) return hipSuccess;
    const unsigned groups = (unsigned)((nChunks + kWavesPerGroup - 1) / kWavesPerGroup);
    if (mantFmt == MRC_MANTISSA_I16)
        hipLaunchKernelGGL((pack_plan_kernel<unsigned short>), dim3(groups), dim3(kWave * kWavesPerGroup), 0, st, S, P, T,
                           nChunks, bitAlloc, (const unsigned short*)mant, tableIn, tableOut, bitsSaved, W.chunkBytes,
                           W.errorFlag, pack_fast16(S, mant, 2, allBandsNonEmpty), chunkMap);
    else
        hipLaunchKernelGGL((pack_plan_kernel<int>), dim3(groups), dim3(kWave * kWavesPerGroup), 0, st, S, P, T, nChunks,
                           bitAlloc, (const int*)mant, tableIn, tableOut, bitsSaved, W.chunkBytes, W.errorFlag,
                           pack_fast16(S, mant, 4, allBandsNonEmpty), chunkMap);
    return hipGetLastError();
}

hipError_t launch_pack_scan(int64_t nChunks, int nch, const PackWs& W, long long* blockOffset, const int* chunkStream,
                            int hdrLen, hipStream_t st) {
    if (nChunks <= 0) return hipSuccess;
    const int64_t nTiles = (nChunks + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(pack_scan_sums_kernel, dim3((unsigned)nTiles), dim3(256), 0, st, nChunks, W.chunkBytes, W.tileSum);
    hipLaunchKernelGGL(pack_scan_tiles_kernel, dim3(1), dim3(64), 0, st, (int)nTiles, W.tileSum, W.total);
    hipLaunchKernelGGL(pack_scan_apply_kernel, dim3((unsigned)nTiles), dim3(256), 0, st, nChunks, nch, W.chunkBytes,
                       W.tileSum, W.total, W.pos, blockOffset, chunkStream, hdrLen);
    hipLaunchKernelGGL(pack_scan_finish_kernel, dim3(1), dim3(64), 0, st, nChunks, W.total, W.pos, chunkStream, hdrLen);
    return hipGetLastError();
}

hipError_t launch_pack_write(const DevShape& S, const PackParams& P, const PackTables& T, int64_t nBlocks,
                             const int* oscale, const int* msSwitch, const int* scaleFactor, const int* bitAlloc,
                             const void* mant, int mantFmt, const int* table, const PackWs& W, const long long* chunkMap,
                             unsigned char* out, long long outCap, int boundBytes, bool allBandsNonEmpty, hipStream_t st) {
    const int64_t nChunks = nBlocks * P.nch;
    if (nChunks <= 0) return hipSuccess;
    const unsigned groups = (unsigned)((nChunks + kWavesPerGroup - 1) / kWavesPerGroup);
    const int wordsPerWave = (boundBytes + 3) / 4 + 2;
    const int prefLen = (S.halfN + (S.halfN >> 4)) + 2;
    const size_t lds = sizeof(unsigned) * (4 * (kPackLutSize + 1) + (size_t)kWavesPerGroup * (wordsPerWave + prefLen + kWave));
    if (mantFmt == MRC_MANTISSA_I16)
        hipLaunchKernelGGL((pack_write_kernel<unsigned short>), dim3(groups), dim3(kWave * kWavesPerGroup), lds, st, S, P,
                           T, nChunks, oscale, msSwitch, scaleFactor, bitAlloc, (const unsigned short*)mant, table,
                           W.chunkBytes, W.pos, out, outCap, wordsPerWave, pack_fast16(S, mant, 2, allBandsNonEmpty),
                           chunkMap, W.errorFlag);
    else
        hipLaunchKernelGGL((pack_write_kernel<int>), dim3(groups), dim3(kWave * kWavesPerGroup), lds, st, S, P, T, nChunks,
                           oscale, msSwitch, scaleFactor, bitAlloc, (const int*)mant, table, W.chunkBytes, W.pos, out,
                           outCap, wordsPerWave, pack_fast16(S, mant, 4, allBandsNonEmpty), chunkMap, W.errorFlag);
    return hipGetLastError();
}

hipError_t launch_pack(const DevShape& S, const PackParams& P, const PackTables& T, int64_t nBlocks, const int* oscale,
                       const int* msSwitch, const int* scaleFactor, const int* bitAlloc, const void* mant, int mantFmt,
                       const int* tableIn, int* tableOut, int* bitsSaved, unsigned char* out, long long outCap,
                       long long* blockOffset, void* ws, int boundBytes, bool allBandsNonEmpty, hipStream_t st) {
    if (nBlocks <= 0) return hipSuccess;
    const int64_t nChunks = nBlocks * P.nch;
    const PackWs W = pack_ws_views(ws, nChunks);
    (void)hipMemsetAsync(W.errorFlag, 0, sizeof(int), st);
    hipError_t e = launch_pack_plan(S, P, T, nBlocks, bitAlloc, mant, mantFmt, tableIn, tableOut, bitsSaved, W, nullptr,
                                    allBandsNonEmpty, st);
    if (e != hipSuccess) return e;
    if ((e = launch_pack_scan(nChunks, P.nch, W, blockOffset, nullptr, 0, st)) != hipSuccess) return e;
    return launch_pack_write(S, P, T, nBlocks, oscale, msSwitch, scaleFactor, bitAlloc, mant, mantFmt, tableOut, W, nullptr,
                             out, outCap, boundBytes, allBandsNonEmpty, st);
}

const int* pack_error_flag(const void* ws, int64_t nChunks) { return pack_ws_views(const_cast<void*>(ws), nChunks).errorFlag; }
const long long* pack_total_bytes(const void* ws, int64_t nChunks) { return pack_ws_views(const_cast<void*>(ws), nChunks).total; }

}  // namespace mrc
