// Chained stream encode on gfx950: the back end of the reference's encode loop (pacfileThem.py:1159-1214 + Close(),
// 973-984) for whole streams, with the bit reservoir carried from block to block ON THE DEVICE.
//
// The reference encodes one block at a time because block t+1's bit budget contains what block t left over
// (codecThem.py:224,274,332,503).  Only the bit allocation and what follows it depend on that number; the transform,
// the psychoacoustic model and the M/S decision do not (phase A: the batch kernels, one launch set per block shape over
// ALL blocks of all streams).  This file is the rest:
//
//   chain_prep_kernel     per block, still reservoir-free: (a) the lines each coded stream will quantise (Mid-or-Left,
//                         Side-or-Right per band, ms_stereo.py:70-81 / codecThem.py:524-551), scaled by their overall
//                         scale (codecThem.py:323, exact), and the per-band peaks likewise; (b) bitalloc.py:106-155's
//                         greedy loop unrolled into its SORTED LIST OF GRANT EVENTS.  The loop always serves the band
//                         with the largest running SMR (first index wins ties), and a band's running SMR only ever
//                         falls (-12 for its first grant, which gives two bits, -6 per further bit): the order in which
//                         grants are ATTEMPTED is the merge of the per-band key sequences -- independent of the budget.
//                         The keys are computed by the loop's own subtractions; the candidate order comes from the
//                         6 dB periodicity (level = floor(SMR / 6) - grants, position inside a level by SMR mod 6) and
//                         is then CHECKED against the actual keys pair by pair (a near-tie that rounding turned round
//                         is put right by an insertion pass), so the list is exactly np.argmax's order;
//   chain_phase_b_kernel  one workgroup per stream walks the stream's blocks in file order: budget from the reservoir
//                         (codecThem.py:299-308, 381-396) -> how far down the event list the budget reaches (all
//                         grants up to the point where fewer than max(nLines) bits are left fit for certain: one
//                         parallel count over the list's cost prefix sums; the few events after it are walked one by
//                         one with the loop's own tests, incl. the 2-bit grant's under-check and retirement) -> scale
//                         factors and mantissas (quantize.py:114-146, 294-322) -> price of the four Huffman tables
//                         (codecThem.py:136-180) -> next reservoir (codecThem.py:224,274).  No host round trip, no
//                         kernel launch per block.
//   chain_flush_gather_kernel, chain_header_kernel: Close()'s block (last hop + zeros) and the file headers.
//
// The budget arithmetic is integer where the reference's is float: bitsLeft = budget - (an integer) is exact in
// float64 for every value the loop can reach (|bitsLeft| <= |budget|, both multiples of ulp(budget)), so
// `nLines <= bitsLeft` is `nLines + spent <= floor(budget)` and `bitsLeft > 0` is `spent < ceil(budget)`.
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
namespace {

constexpr int kChainThreads = 256;
constexpr int kMaxEvents = 64 * 15;                   // bands (x streams) <= 64, grants per band <= maxMantBits - 1 <= 15
constexpr int kLutSize = 65;                          // largest value in any Huffman table is 64 (index 65: any other)

// code length per value (0 = not in the table); rows: percussive, silence, speech, tonal (sorted names; the table data
// of mrc_pack.cpp / mrc_kernels_huff.hip)
__constant__ unsigned char kChainCodeLen[4][kLutSize] = {
    {1, 4, 3, 6, 3, 4, 6, 8, 5, 6, 7, 7, 7, 9, 9, 0, 6},
    {2, 3, 3, 5, 2, 4, 6, 0, 5, 5, 6, 4},
    {2, 4, 3, 6, 2, 4, 6, 4, 4, 5, 6, 7, 7, 0, 0, 0, 6, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7},
    {1, 5, 3, 7, 3, 7, 8, 4, 4, 7, 8, 0, 0, 0, 0, 0, 5, 7, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6,
     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8}};
__constant__ int kChainEscape[4] = {16, 11, 7, 7};

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
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
    const uint2v r16 = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)r16.x, (int)r16.y);
    const uint2v r32 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)r32.x, (int)r32.y);
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
// LDS traffic between the lanes of ONE wave
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// the order np.argmax serves grant attempts in: larger key first, equal keys by band index
__device__ __forceinline__ bool event_before(double ka, int ba, double kb, int bb) {
    return ka > kb || (ka == kb && ba < bb);
}

// ------------------------------------------------------------------------------------------------------------------
// prep: one workgroup per block of one shape group
// ------------------------------------------------------------------------------------------------------------------
// event record: band | bitsAfter << 6 | nLines << 11 (nLines <= 2^20)
__global__ __launch_bounds__(kChainThreads) void chain_prep_kernel(
    DevShape S, int joint, int64_t nBlocks, const double* __restrict__ lines, const int* __restrict__ oscale,
    const double* __restrict__ smr, const double* __restrict__ peak, const int* __restrict__ msSwitch,
    double* __restrict__ xsel, double* __restrict__ peakSel, unsigned* __restrict__ evOut, unsigned* __restrict__ preOut,
    unsigned short* __restrict__ posOut, int forceFallback /* tests: scramble the candidate order first */) {
    __shared__ unsigned sEv[kMaxEvents];
    __shared__ unsigned sPre[kMaxEvents + 1];
    __shared__ double sKey[kMaxEvents];
    __shared__ unsigned short sPos[kMaxEvents];
    __shared__ double sPhi[kWave], sSmr[kWave];
    __shared__ int sQ[kWave], sSlot[kWave];
    __shared__ int sSig[2 * kMaxBands];                  // signal of (stream, band)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int64_t blk = blockIdx.x;
    const int nb = S.nBands, M = S.halfN;
    const int nsig = joint ? 4 : 1, nstream = joint ? 2 : 1, nTot = nstream * nb;
    const int K = S.maxMantBits - 1;
    const int nEv = nTot * K;
    const int* osc = oscale + blk * nsig;
    if (tid < nTot) {
        const int band = tid % nb, strm = tid / nb;
        const int sig = joint ? (msSwitch[blk * nb + band] ? 2 + strm : strm) : 0;   // ms_stereo.py:70-81
        sSig[tid] = sig;
        // codecThem.py:346-347: the band's scale factor comes from max |scaled line|; scaling by 2^overallScale is exact
        peakSel[blk * nTot + tid] = ldexp(peak[(blk * nsig + sig) * nb + band], osc[sig]);
    }
    __syncthreads();
    // (a) the lines of the coded streams, scaled
    for (int u = tid; u < nstream * M; u += kChainThreads) {
        const int strm = u / M, k = u - strm * M;
        const int sig = sSig[strm * nb + S.bandOfLine[k]];
        xsel[(blk * nstream + strm) * (int64_t)M + k] = ldexp(lines[(blk * nsig + sig) * (int64_t)M + k], osc[sig]);
    }
    if (wave != 0) return;
    // (b) the sorted grant events.  lane i < nTot = (stream, band) i of bitalloc.py's concatenated arrays
    // (codecThem.py:491-498)
    const bool valid = lane < nTot;
    double s = 0.0;
    if (valid) s = smr[(blk * nsig + sSig[lane]) * nb + lane % nb];
    {
        double qd = floor(s / 6.0);
        qd = fmin(fmax(qd, -1000000.0), 1000000.0);         // (garbage in: still a bounded, valid candidate order)
        if (!(qd == qd)) qd = 0.0;
        sQ[lane] = (int)qd;
        sPhi[lane] = s - 6.0 * qd;
        sSmr[lane] = s;
    }
    wave_sync();
    // position of the band inside a 6 dB level: by SMR mod 6, larger first, equal ones by index
    int rank = 0;
    {
        const double myPhi = sPhi[lane];
        for (int j = 0; j < nTot; ++j) {
            const double pj = sPhi[j];
            rank += (pj > myPhi || (pj == myPhi && j < lane)) ? 1 : 0;
        }
    }
    if (valid) sSlot[rank] = lane;
    wave_sync();
    // from here on lane r holds the band with rank r
    const int band = valid ? sSlot[lane] : 0;
    const int nB = valid ? S.bandN[band % nb] : 0;
    const int qB = sQ[band];
    double cur = sSmr[band];                               // the band's running SMR (bitalloc.py:139,146)
    int k = 0;
    int nextLevel = (valid && K > 0) ? qB : INT_MIN;
    int base = 0, costBase = 0;
    for (int guard = 0; guard <= kMaxEvents; ++guard) {    // every pass emits at least one event
        const int level = wave_max_i(nextLevel);
        if (level == INT_MIN) break;
        const bool has = nextLevel == level;
        const unsigned long long mask = __ballot(has);
        const int off = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        const int cost = has ? (k == 0 ? 2 * nB : nB) : 0;  // bitalloc.py:137-138, 144-145
        const int incl = wave_scan_i(cost);
        if (has) {
            const int p = base + off;
            sEv[p] = (unsigned)band | ((unsigned)(k + 2) << 6) | ((unsigned)nB << 11);
            sPre[p] = (unsigned)(costBase + incl - cost);
            sKey[p] = cur;
            sPos[k * nTot + band] = (unsigned short)p;
            cur -= (k == 0) ? 12.0 : 6.0;
            ++k;
            nextLevel = k >= K ? INT_MIN : (k == 1 ? qB - 2 : nextLevel - 1);
        }
        base += __popcll(mask);
        costBase += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) sPre[nEv] = (unsigned)costBase;
    wave_sync();
    // the candidate order against the keys themselves
    bool bad = false;
    for (int p = lane; p + 1 < nEv; p += kWave)
        bad |= !event_before(sKey[p], (int)(sEv[p] & 63u), sKey[p + 1], (int)(sEv[p + 1] & 63u));
    if (forceFallback) {
        // tests: exchange neighbouring events so that the repair below has real work
        wave_sync();
        for (int p = 2 * lane; p + 1 < nEv; p += 2 * kWave) {
            const unsigned e0 = sEv[p]; sEv[p] = sEv[p + 1]; sEv[p + 1] = e0;
            const double k0 = sKey[p]; sKey[p] = sKey[p + 1]; sKey[p + 1] = k0;
        }
        bad = true;
    }
    if (__any(bad)) {
        wave_sync();
        if (lane == 0) {
            for (int i = 1; i < nEv; ++i) {                 // insertion pass: the list is sorted but for a few neighbours
                const unsigned e = sEv[i];
                const double key = sKey[i];
                int j = i;
                while (j > 0 && event_before(key, (int)(e & 63u), sKey[j - 1], (int)(sEv[j - 1] & 63u))) {
                    sEv[j] = sEv[j - 1]; sKey[j] = sKey[j - 1]; --j;
                }
                sEv[j] = e; sKey[j] = key;
            }
            unsigned run = 0;
            for (int p = 0; p < nEv; ++p) {
                const unsigned e = sEv[p];
                const int bnd = (int)(e & 63u), after = (int)((e >> 6) & 31u), nn = (int)(e >> 11);
                sPre[p] = run;
                sPos[(after - 2) * nTot + bnd] = (unsigned short)p;
                run += (unsigned)(after == 2 ? 2 * nn : nn);
            }
            sPre[nEv] = run;
        }
        wave_sync();
    }
    unsigned* ev = evOut + blk * (int64_t)nEv;
    unsigned* pre = preOut + blk * (int64_t)(nEv + 1);
    unsigned short* pos = posOut + blk * (int64_t)nEv;
    for (int p = lane; p < nEv; p += kWave) { ev[p] = sEv[p]; pos[p] = sPos[p]; }
    for (int p = lane; p <= nEv; p += kWave) pre[p] = sPre[p];
}

// ------------------------------------------------------------------------------------------------------------------
// phase B: one workgroup per stream
// ------------------------------------------------------------------------------------------------------------------
#ifdef MRC_CHAIN_PROFILE
// profiling build only (make EXTRA=-DMRC_CHAIN_PROFILE): shader-clock cycles of wave 0 per phase of chain_phase_b_kernel
__device__ unsigned long long gChainProf[16];
#define MRC_CP(i) do { const long long now_ = clock64(); if (tid == 0) atomicAdd(&gChainProf[i], (unsigned long long)(now_ - tProf_)); tProf_ = clock64(); } while (0)
#else
#define MRC_CP(i) do { } while (0)
#endif
// What one thread holds of an item between its loads and its use: the loads of item t + 1 are issued BEFORE item t is
// computed and land while it runs (each item's inputs are cold in HBM; without this every block would start by waiting
// ~2 us for them).
// NT threads per workgroup: 256 when many streams share the chip (eight streams per CU), 512 for a few long streams (the
// line-parallel half of an item -- mantissas, prices -- then runs on eight waves instead of four).
constexpr int kMaxLinesPerItem = 2 * 1024;                                    // two coded streams of <= 1024 lines
template <int NT> struct ChainDims {
    static constexpr int kEvPerThread = (kMaxEvents + NT) / NT;               // the events / prefix sums a thread stages
    static constexpr int kUnitsPerThread = (kMaxLinesPerItem / 4 + NT - 1) / NT;   // units of four lines per thread
};
constexpr int kEvPerLane = kMaxEvents / kWave;                                // 15
constexpr int kMaxGrants = 15;                                                // maxMantBits - 1 <= 15
template <int NT> struct ItemRegs {
    unsigned ev[ChainDims<NT>::kEvPerThread], pre[ChainDims<NT>::kEvPerThread];
    unsigned short pos[ChainDims<NT>::kEvPerThread];
    double peak;
    int bandN;
    double2 xa[ChainDims<NT>::kUnitsPerThread], xb[ChainDims<NT>::kUnitsPerThread];
    unsigned bands[ChainDims<NT>::kUnitsPerThread];
};
// The fields of one group descriptor, held in scalar registers: they are read once per CHANGE of block shape (a stream is
// mostly runs of long blocks), not as dependent scalar loads inside every item.  The empty asm makes each value opaque,
// so the compiler keeps it in its register instead of loading it again from the descriptor where it is used.
struct GroupView {
    int joint, nb, nTot, M, K, nEv, maxN, nScaleBits, nstream;
    double budgetMono, budgetJointPre, blkswA, blkswB;
    const unsigned char* bandOfLine;
    const int* bandN;
    const double* xsel;
    const double* peakSel;
    const unsigned* ev;
    const unsigned* pre;
    const unsigned short* pos;
    int* bitAlloc;
    int* scaleFactor;
    unsigned short* mant;
    int* table;
};
#define MRC_PIN(x) asm volatile("" : "+s"(x))
__device__ __forceinline__ GroupView group_view(const ChainGroupDev* __restrict__ groups, int g) {
    const ChainGroupDev& D = groups[g];
    GroupView V;
    V.joint = D.joint; V.nb = D.nb; V.nTot = D.nTot; V.M = D.M; V.K = D.K; V.nEv = D.nEv; V.maxN = D.maxN;
    V.nScaleBits = D.nScaleBits; V.nstream = D.nstream;
    V.budgetMono = D.budgetMono; V.budgetJointPre = D.budgetJointPre; V.blkswA = D.blkswA; V.blkswB = D.blkswB;
    V.bandOfLine = D.bandOfLine; V.bandN = D.bandN; V.xsel = D.xsel; V.peakSel = D.peakSel; V.ev = D.ev; V.pre = D.pre;
    V.pos = D.pos; V.bitAlloc = D.bitAlloc; V.scaleFactor = D.scaleFactor; V.mant = D.mant; V.table = D.table;
    MRC_PIN(V.joint); MRC_PIN(V.nb); MRC_PIN(V.nTot); MRC_PIN(V.M); MRC_PIN(V.K); MRC_PIN(V.nEv); MRC_PIN(V.maxN);
    MRC_PIN(V.nScaleBits); MRC_PIN(V.nstream);
    MRC_PIN(V.budgetMono); MRC_PIN(V.budgetJointPre); MRC_PIN(V.blkswA); MRC_PIN(V.blkswB);
    MRC_PIN(V.bandOfLine); MRC_PIN(V.bandN); MRC_PIN(V.xsel); MRC_PIN(V.peakSel); MRC_PIN(V.ev); MRC_PIN(V.pre);
    MRC_PIN(V.pos); MRC_PIN(V.bitAlloc); MRC_PIN(V.scaleFactor); MRC_PIN(V.mant); MRC_PIN(V.table);
    return V;
}
template <int NT>
__device__ __forceinline__ void item_load(const GroupView& G, int64_t idx, int tid, ItemRegs<NT>& R) {
    constexpr int kEvPerThread = ChainDims<NT>::kEvPerThread, kUnitsPerThread = ChainDims<NT>::kUnitsPerThread;
    constexpr int kChainThreads = NT;
    const int nEv = G.nEv, nTot = G.nTot, M = G.M, nstream = G.nstream;
    const unsigned* ev = G.ev + idx * (int64_t)nEv;
    const unsigned* pre = G.pre + idx * (int64_t)(nEv + 1);
    const unsigned short* pos = G.pos + idx * (int64_t)nEv;
#pragma unroll
    for (int j = 0; j < kEvPerThread; ++j) {
        const int p = tid + kChainThreads * j;
        R.ev[j] = p < nEv ? ev[p] : 0u;
        R.pos[j] = p < nEv ? pos[p] : (unsigned short)0;
        R.pre[j] = p <= nEv ? pre[p] : 0u;
    }
    R.peak = tid < nTot ? G.peakSel[idx * nTot + tid] : 0.0;
    R.bandN = tid < nTot ? G.bandN[tid >= G.nb ? tid - G.nb : tid] : 0;
    const int upl = M >> 2, nUnits = nstream * upl;
#pragma unroll
    for (int j = 0; j < kUnitsPerThread; ++j) {
        const int u = tid + kChainThreads * j;
        R.xa[j] = make_double2(0.0, 0.0); R.xb[j] = make_double2(0.0, 0.0); R.bands[j] = 0u;
        if (u < nUnits) {
            const int strm = u >= upl ? 1 : 0;
            const int k = 4 * (u - strm * upl);
            const double* src = G.xsel + (idx * nstream + strm) * (int64_t)M + k;
            R.xa[j] = *reinterpret_cast<const double2*>(src);
            R.xb[j] = *reinterpret_cast<const double2*>(src + 2);
            R.bands[j] = *reinterpret_cast<const unsigned*>(G.bandOfLine + k);
        }
    }
}

// quantize.py:12-38 for code widths of at most 31 bits (the encoder's: 2^nScaleBits - 1 + bits <= 15 + 16): the same
// expression as dev::mag_code, the truncation done by the 32-bit conversion instead of the emulated 64-bit one
__device__ __forceinline__ unsigned mag_code32(double mag, int nBits) {
    if (mag >= 1.0) return (1u << (nBits - 1)) - 1u;
    return (unsigned)((((double)((1u << nBits) - 1u)) * mag + 1.0) / 2.0);
}
__device__ __forceinline__ int scale_factor32(double v, int nScaleBits, int nMantBits) {      // quantize.py:114-146
    const int cap = (1 << nScaleBits) - 1;
    const int nBits = cap + nMantBits;
    const unsigned code = mag_code32(fabs(v), nBits);
    const int top = code ? 31 - __clz((int)code) : 0;
    const int lz = (nBits - 2) - top;
    return lz < cap ? lz : cap;
}
__device__ __forceinline__ unsigned mantissa32(double x, int scale, int nScaleBits, int nMantBits) {   // quantize.py:294-322
    const int cap = (1 << nScaleBits) - 1;
    const unsigned code = mag_code32(fabs(x), cap + nMantBits);
    const int shift = cap - scale;
    return (x < 0.0 ? (1u << (nMantBits - 1)) : 0u) + (code >> (shift < 0 ? 0 : shift));
}

template <int NT>
__global__ __launch_bounds__(NT) void chain_phase_b_kernel(
    const ChainGroupDev* __restrict__ groups, const int* __restrict__ items, const long long* __restrict__ itemStart,
    int* __restrict__ reservoir, int* __restrict__ resTrace /* nullable: reservoir after every item */, int useHuffman) {
    constexpr int kEvPerThread = ChainDims<NT>::kEvPerThread, kUnitsPerThread = ChainDims<NT>::kUnitsPerThread;
    constexpr int kChainThreads = NT;
    __shared__ int sBits[kWave];                         // bits granted per (stream, band) while the tail is walked
    __shared__ unsigned sEv[kMaxEvents + kChainThreads];
    __shared__ unsigned sPre[kMaxEvents + kChainThreads];
    __shared__ unsigned short sPos[kMaxEvents + kChainThreads];
    __shared__ double sPeak[kWave];
    __shared__ int sBandN[kWave];
    __shared__ unsigned sInfo[kWave];                    // per (stream, band): bits | scale factor << 8
    __shared__ unsigned sEsc[kWave];                     // per (stream, band): bits + escape code length of each table, 8 bits each
    __shared__ unsigned sLut[kLutSize + 1];              // per value: the four code lengths, 8 bits each (0: not in the table)
    __shared__ unsigned sRed[kChainThreads / kWave][4];
    __shared__ int sCtl[4];                              // remaining bits, raw bits of stream 0 / 1, reservoir
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int64_t strmId = blockIdx.x;
    for (int v = tid; v <= kLutSize; v += kChainThreads) {
        unsigned e = 0;
        if (v < kLutSize)
            for (int t = 0; t < 4; ++t) e |= (unsigned)kChainCodeLen[t][v] << (8 * t);
        sLut[v] = e;
    }
    unsigned escLen4 = 0;                                // the escape code's length in each table
    for (int t = 0; t < 4; ++t) escLen4 |= (unsigned)kChainCodeLen[t][kChainEscape[t]] << (8 * t);
    if (tid == 0) sCtl[3] = reservoir[strmId];
    const long long i0 = itemStart[strmId], i1 = itemStart[strmId + 1];
    ItemRegs<NT> R;
    int itemCur = 0, itemNext = 0;
    GroupView G = group_view(groups, 0), Gn = G;         // the current item's group, the next item's
    int gOfG = 0, gOfGn = 0;
    if (i0 < i1) {
        itemCur = items[i0];
        itemNext = i0 + 1 < i1 ? items[i0 + 1] : 0;
        gOfG = (int)((unsigned)itemCur >> 28);
        if (gOfG != 0) G = group_view(groups, gOfG);
        Gn = G; gOfGn = gOfG;
        item_load(G, itemCur & 0x0fffffff, tid, R);
    }
    __syncthreads();
#ifdef MRC_CHAIN_PROFILE
    long long tProf_ = clock64();
#endif
    for (long long it = i0; it < i1; ++it) {
        const int item = itemCur;
        const int64_t idx = item & 0x0fffffff;
        const int nb = G.nb, nTot = G.nTot, M = G.M, K = G.K, nEv = G.nEv, nstream = G.nstream;
        MRC_CP(0);
        // ---- this item's event list, peaks and band sizes: registers -> LDS; its lines stay in registers
#pragma unroll
        for (int j = 0; j < kEvPerThread; ++j) {
            const int p = tid + kChainThreads * j;
            sEv[p] = R.ev[j]; sPre[p] = R.pre[j]; sPos[p] = R.pos[j];
        }
        if (tid < kWave) { sPeak[tid] = R.peak; sBandN[tid] = R.bandN; }
        double2 xa[kUnitsPerThread], xb[kUnitsPerThread];
        unsigned bandsOf[kUnitsPerThread];
#pragma unroll
        for (int j = 0; j < kUnitsPerThread; ++j) { xa[j] = R.xa[j]; xb[j] = R.xb[j]; bandsOf[j] = R.bands[j]; }
        MRC_CP(1);
        __syncthreads();
        MRC_CP(2);
        // ---- the next item's loads go out now and land while this one is computed (its id came with the previous one)
        const int itemAfter = it + 2 < i1 ? items[it + 2] : 0;
        if (it + 1 < i1) {
            const int gn = (int)((unsigned)itemNext >> 28);
            if (gn != gOfGn) { Gn = group_view(groups, gn); gOfGn = gn; }    // (a change of block shape: rare)
            item_load(Gn, itemNext & 0x0fffffff, tid, R);
        }
        MRC_CP(3);
        if (wave == 0) {
            // ---- bit allocation (bitalloc.py:106-155) for the budget of codecThem.py:299-308 / 381-396
            const double r = (double)sCtl[3];
            double budget;
            if (G.joint) { budget = G.budgetJointPre + r; budget -= G.blkswA; budget -= G.blkswB; }
            else budget = G.budgetMono + r;
            // nLines <= left  <=>  nLines + spent <= Bf;  left > 0  <=>  spent < Bc  (spent: an integer below 2^16)
            const int Bf = (int)fmin(fmax(floor(budget), -1.0e9), 1.0e9), Bc = (int)fmin(fmax(ceil(budget), -1.0e9), 1.0e9);
            // how many events are certain grants: fewer than maxN bits have been spent short of the budget
            const int maxN = G.maxN;
            // (all reads unconditional and issued together -- the arrays are padded -- then masked: a guarded read would wait
            // for its LDS round trip before the next one is issued)
            int preV[kEvPerLane];
#pragma unroll
            for (int j = 0; j < kEvPerLane; ++j) preV[j] = (int)sPre[lane + kWave * j];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < kEvPerLane; ++j) cnt += ((lane + kWave * j < nEv) & (preV[j] + maxN <= Bf)) ? 1 : 0;
            const int cut = wave_sum_i(cnt);
            const bool valid = lane < nTot;
            int posV[kMaxGrants];
#pragma unroll
            for (int k = 0; k < kMaxGrants; ++k) posV[k] = (int)sPos[min(k * nTot, kMaxEvents) + lane];
            int c = 0;
#pragma unroll
            for (int k = 0; k < kMaxGrants; ++k) c += ((k < K) & (posV[k] < cut)) ? 1 : 0;
            int myBits = (valid && c) ? c + 1 : 0;                         // the first grant gives two bits
            const int myN = valid ? sBandN[lane] : 0;
            int spent = (int)sPre[cut];
            MRC_CP(4);
            // The tail, in batches: the next 64 events at a time.  A band that no longer fits can never be granted again
            // (the bits left only shrink), so it is retired at once -- as bitalloc.py:149-151 would retire it when it next
            // came up -- and every event of a live band is a grant UNLESS the grants of live events before it in the batch
            // have used its room up: a prefix sum over the batch's live costs finds the first such event; everything before
            // it is granted in one go, the live set is brought up to date, and the rest of the batch is looked at again.
            unsigned long long alive = nTot >= 64 ? ~0ull : ((1ull << nTot) - 1ull);
            alive &= ~__ballot(valid && myN + spent > Bf);
            sBits[lane] = myBits;
            int e = cut;
            bool done = !(spent < Bc) || alive == 0ull;
            while (!done && e < nEv) {
                const bool evValid = e + lane < nEv;
                const unsigned rec = evValid ? sEv[e + lane] : 0u;
                const int evBand = (int)(rec & 63u), evN = (int)(rec >> 11), evAfter = (int)((rec >> 6) & 31u);
                const int evCost = evAfter == 2 ? 2 * evN : evN;                             // bitalloc.py:137-138, 144-145
                unsigned long long pending = __ballot(evValid);
                while (pending != 0ull && !done) {
                    const bool live = ((pending >> lane) & 1ull) && ((alive >> evBand) & 1ull);
                    const int cst = live ? evCost : 0;
                    const int incl = wave_scan_i(cst);
                    const int before = spent + incl - cst;
                    // bitalloc.py:131,134: the loop goes on while bits are left, and only nLines is tested (also for the
                    // two-bit grant)
                    const bool ok = !live || (evN + before <= Bf && before < Bc);
                    const unsigned long long bad = __ballot(!ok);
                    const int f = bad ? __builtin_ctzll(bad) : kWave;
                    if (live && lane < f) atomicMax(&sBits[evBand], evAfter);               // (a band's later grants carry more bits)
                    spent += f ? __builtin_amdgcn_readlane(incl, f - 1) : 0;
                    pending = f >= kWave - 1 ? 0ull : (pending & ~((2ull << f) - 1ull));     // event f itself: retired below, or the end
                    alive &= ~__ballot(valid && myN + spent > Bf);
                    done = !(spent < Bc) || alive == 0ull;
                }
                e += kWave;
            }
            wave_sync();
            myBits = sBits[lane];
            MRC_CP(5);
#ifdef MRC_CHAIN_PROFILE
            if (tid == 0) { atomicAdd(&gChainProf[12], (unsigned long long)(e - cut)); atomicAdd(&gChainProf[13], 1ull); }
#endif
            // ---- scale factors (codecThem.py:346-347), raw size of each stream (codecThem.py:141-146)
            const int rawMine = myBits * myN;
            const int raw0 = wave_sum_i((valid && lane < nb) ? rawMine : 0);
            const int raw1 = wave_sum_i((valid && lane >= nb) ? rawMine : 0);
            if (valid) {
                const int sf = scale_factor32(sPeak[lane], G.nScaleBits, myBits);
                sInfo[lane] = (unsigned)myBits | ((unsigned)sf << 8);
                sEsc[lane] = escLen4 + (unsigned)myBits * 0x01010101u;
                G.bitAlloc[idx * nTot + lane] = myBits;
                G.scaleFactor[idx * nTot + lane] = sf;
            }
            if (lane == 0) {
                sCtl[0] = (int)(budget - (double)spent);                   // int(bitsLeft): truncation toward zero (bitalloc.py:155)
                sCtl[1] = raw0;
                sCtl[2] = raw1;
            }
            MRC_CP(6);
        }
        __syncthreads();
        MRC_CP(7);
        // ---- mantissas (codecThem.py:348-349) and the price of every Huffman table (codecThem.py:157-173): the four
        //      prices of a line are four bytes of one word (at most 22 each, at most eight lines per thread)
        unsigned accA = 0u, accB = 0u;
        {
            const int upl = M >> 2;                                        // units of four lines per stream
            const int nUnits = nstream * upl;
            const int nScaleBits = G.nScaleBits;
#pragma unroll
            for (int jj = 0; jj < kUnitsPerThread; ++jj) {
                const int u = tid + kChainThreads * jj;
                if (u < nUnits) {
                    const int strm = u >= upl ? 1 : 0;
                    const int k = 4 * (u - strm * upl);
                    const unsigned bands = bandsOf[jj];
                    const double x[4] = {xa[jj].x, xa[jj].y, xb[jj].x, xb[jj].y};
                    unsigned code[4], acc = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int sb = strm * nb + (int)((bands >> (8 * j)) & 0xffu);
                        const unsigned info = sInfo[sb];
                        const int ba = (int)(info & 0xffu);
                        unsigned cdv = 0u;
                        if (ba) {
                            cdv = mantissa32(x[j], (int)(info >> 8), nScaleBits, ba);
                            const unsigned lens = sLut[cdv < (unsigned)kLutSize ? cdv : (unsigned)kLutSize];
                            // a value without a code (length byte 0) costs the escape code + the raw mantissa; the escape
                            // VALUE itself is priced as its code alone (codecThem.py:169-172)
                            unsigned z = (lens & 0x7f7f7f7fu) + 0x7f7f7f7fu;
                            z = ~(z | lens | 0x7f7f7f7fu);                 // 0x80 in exactly the zero bytes
                            const unsigned msk = (z >> 7) * 0xffu;
                            acc += (lens & ~msk) | (sEsc[sb] & msk);
                        }
                        code[j] = cdv;
                    }
                    if (strm) accB += acc; else accA += acc;
                    uint2 w;
                    w.x = code[0] | (code[1] << 16);
                    w.y = code[2] | (code[3] << 16);
                    *reinterpret_cast<uint2*>(G.mant + (idx * nstream + strm) * (int64_t)M + k) = w;
                }
            }
        }
        {
            // bytes -> 16-bit fields (per table at most 25 bits x 1024 lines): tables 0 | 2 and 1 | 3 of each stream
            const unsigned w0 = (unsigned)wave_sum_i((int)(accA & 0x00ff00ffu)), w1 = (unsigned)wave_sum_i((int)((accA >> 8) & 0x00ff00ffu));
            const unsigned w2 = (unsigned)wave_sum_i((int)(accB & 0x00ff00ffu)), w3 = (unsigned)wave_sum_i((int)((accB >> 8) & 0x00ff00ffu));
            if (lane == 0) { sRed[wave][0] = w0; sRed[wave][1] = w1; sRed[wave][2] = w2; sRed[wave][3] = w3; }
        }
        MRC_CP(8);
        __syncthreads();
        MRC_CP(9);
        if (tid == 0) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int v = 0; v < kChainThreads / kWave; ++v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] += sRed[v][q];
            }
            int res = sCtl[0];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s < nstream) {
                    const int raw = sCtl[1 + s];
                    const int cost[4] = {(int)(w[2 * s] & 0xffffu), (int)(w[2 * s + 1] & 0xffffu), (int)(w[2 * s] >> 16),
                                         (int)(w[2 * s + 1] >> 16)};
                    int best = raw, table = 15;                           // codecThem.py:147-149
                    if (useHuffman) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (cost[t] < best) { best = cost[t]; table = t; }   // strictly less: raw, then the first table, win ties
                    }
                    G.table[idx * nstream + s] = table;
                    res += raw - best;                                    // codecThem.py:202,224,274
                }
            }
            sCtl[3] = res;
            if (resTrace) resTrace[it] = res;
        }
        itemCur = itemNext;
        itemNext = itemAfter;
        G = Gn; gOfG = gOfGn;
        MRC_CP(10);
        __syncthreads();
        MRC_CP(11);
    }
    if (tid == 0) reservoir[strmId] = sCtl[3];
}

// Close() (pacfileThem.py:973-984): per stream and channel the last coded hop followed by a hop of zeros
template <class T>
__global__ void chain_flush_gather_kernel(int64_t nStreams, int L, const T* __restrict__ pcmL, const T* __restrict__ pcmR,
                                          int64_t stride, const long long* __restrict__ tailOffset, T* __restrict__ out) {
    const int64_t u = blockIdx.x;                         // stream * 2 + channel
    const int64_t s = u >> 1;
    const T* src = ((u & 1) ? pcmR : pcmL) + s * stride + tailOffset[s];
    T* dst = out + u * 2 * (int64_t)L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) { dst[i] = src[i]; dst[L + i] = (T)0; }
}

// the file header of every stream in front of its first chunk
__global__ void chain_header_kernel(int64_t nStreams, int hdrLen, const unsigned char* __restrict__ hdr,
                                    const long long* __restrict__ firstChunk, const long long* __restrict__ pos,
                                    unsigned char* __restrict__ out, long long outCap) {
    const int64_t s = blockIdx.x;
    const long long p0 = pos[firstChunk[s]] - hdrLen;
    if (p0 < 0 || p0 + hdrLen > outCap) return;
    for (int i = threadIdx.x; i < hdrLen; i += blockDim.x) out[p0 + i] = hdr[s * hdrLen + i];
}

}  // namespace

#ifdef MRC_CHAIN_PROFILE
extern "C" int mrc_debug_chain_profile(unsigned long long* out /*[16]*/, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gChainProf), sizeof(gChainProf)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gChainProf), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif

size_t chain_events_per_block(const DevShape& S, int joint) { return (size_t)(joint ? 2 : 1) * S.nBands * (S.maxMantBits - 1); }

hipError_t launch_chain_prep(const DevShape& S, int joint, int64_t nBlocks, const double* lines, const int* oscale,
                             const double* smr, const double* peak, const int* msSwitch, double* xsel, double* peakSel,
                             unsigned* ev, unsigned* pre, unsigned short* pos, int forceFallback, hipStream_t st) {
    if (nBlocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_prep_kernel, dim3((unsigned)nBlocks), dim3(kChainThreads), 0, st, S, joint, nBlocks, lines,
                       oscale, smr, peak, msSwitch, xsel, peakSel, ev, pre, pos, forceFallback);
    return hipGetLastError();
}

hipError_t launch_chain_phase_b(int64_t nStreams, const ChainGroupDev* groups, const int* items, const long long* itemStart,
                                int* reservoir, int* resTrace, int useHuffman, int threads, hipStream_t st) {
    if (nStreams <= 0) return hipSuccess;
    // a workgroup per stream.  Few streams: large workgroups (the chip is idle anyway, the stream's latency is what
    // counts); many: small ones, eight streams per CU
    if (threads <= 0) threads = nStreams <= 512 ? 512 : 256;   // (measured on one stream: 4.08 / 3.58 / 3.67 us per block at 256 / 512 / 1024)
    if (threads >= 1024)
        hipLaunchKernelGGL(chain_phase_b_kernel<1024>, dim3((unsigned)nStreams), dim3(1024), 0, st, groups, items, itemStart,
                           reservoir, resTrace, useHuffman);
    else if (threads >= 512)
        hipLaunchKernelGGL(chain_phase_b_kernel<512>, dim3((unsigned)nStreams), dim3(512), 0, st, groups, items, itemStart,
                           reservoir, resTrace, useHuffman);
    else
        hipLaunchKernelGGL(chain_phase_b_kernel<256>, dim3((unsigned)nStreams), dim3(256), 0, st, groups, items, itemStart,
                           reservoir, resTrace, useHuffman);
    return hipGetLastError();
}

hipError_t launch_chain_flush_gather(int64_t nStreams, int L, const void* pcmL, const void* pcmR, int fmt, int64_t stride,
                                     const long long* tailOffset, void* out, hipStream_t st) {
    if (nStreams <= 0) return hipSuccess;
    if (fmt == kSampleI16)
        hipLaunchKernelGGL(chain_flush_gather_kernel<short>, dim3((unsigned)(2 * nStreams)), dim3(256), 0, st, nStreams, L,
                           (const short*)pcmL, (const short*)pcmR, stride, tailOffset, (short*)out);
    else
        hipLaunchKernelGGL(chain_flush_gather_kernel<double>, dim3((unsigned)(2 * nStreams)), dim3(256), 0, st, nStreams, L,
                           (const double*)pcmL, (const double*)pcmR, stride, tailOffset, (double*)out);
    return hipGetLastError();
}

hipError_t launch_chain_headers(int64_t nStreams, int hdrLen, const unsigned char* hdr, const long long* firstChunk,
                                const long long* pos, unsigned char* out, long long outCap, hipStream_t st) {
    if (nStreams <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_header_kernel, dim3((unsigned)nStreams), dim3(64), 0, st, nStreams, hdrLen, hdr, firstChunk,
                       pos, out, outCap);
    return hipGetLastError();
}

}  // namespace mrc
