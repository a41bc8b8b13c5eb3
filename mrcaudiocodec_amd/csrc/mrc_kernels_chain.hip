// Chained stream encode on gfx950: the back end of the reference's encode loop (pacfileThem.py:1159-1214 + Close(),
// 973-984) for whole streams, with the bit reservoir carried from block to block ON THE DEVICE.
//
// The reference encodes one block at a time because block t+1's bit budget contains what block t left over
// (codecThem.py:224,274,332,503).  Only the bit allocation and what follows it depend on that number; the transform,
// the psychoacoustic model and the M/S decision do not (phase A: the batch kernels, one launch set per block shape over
// ALL blocks of all streams).  This file is the rest:
//
//   chain_prep_kernel     per block, still reservoir-free (a wave per block): bitalloc.py:106-155's greedy loop unrolled
//                         into its SORTED LIST OF GRANT EVENTS.  The loop always serves the band
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
// Four per-lane partial sums -> every lane of row r (lanes 16 r .. 16 r + 15) ends with the wave-wide sum of value r: two
// rounds of permlane swaps fold the four registers into one whose rows belong to the four values, then one reduction inside
// the rows (10 instead of 32 instructions for four separate sums)
__device__ __forceinline__ unsigned wave_sum4_u(unsigned a, unsigned b, unsigned c, unsigned d) {
    uint2v t = __builtin_amdgcn_permlane32_swap(a, c, false, false);    // x = {a lower half, c lower half}, y = the upper halves
    const unsigned ac = t.x + t.y;                                      // lanes 0-31: a over both halves; lanes 32-63: c
    t = __builtin_amdgcn_permlane32_swap(b, d, false, false);
    const unsigned bd = t.x + t.y;
    t = __builtin_amdgcn_permlane16_swap(ac, bd, false, false);         // x = {ac row 0, bd row 0, ac row 2, bd row 2}, y = the odd rows
    int q = (int)(t.x + t.y);                                           // row r: value r
    q += __builtin_amdgcn_update_dpp(0, q, 0xB1, 0xf, 0xf, false);
    q += __builtin_amdgcn_update_dpp(0, q, 0x4E, 0xf, 0xf, false);
    q += __builtin_amdgcn_update_dpp(0, q, 0x141, 0xf, 0xf, false);
    q += __builtin_amdgcn_update_dpp(0, q, 0x128, 0xf, 0xf, false);
    return (unsigned)q;
}
// ... two values: lanes 0-31 end with the sum of a, lanes 32-63 with the sum of b
__device__ __forceinline__ int wave_sum2_i(int a, int b) {
    uint2v t = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    int q = (int)(t.x + t.y);
    q += __builtin_amdgcn_update_dpp(0, q, 0xB1, 0xf, 0xf, false);
    q += __builtin_amdgcn_update_dpp(0, q, 0x4E, 0xf, 0xf, false);
    q += __builtin_amdgcn_update_dpp(0, q, 0x141, 0xf, 0xf, false);
    q += __builtin_amdgcn_update_dpp(0, q, 0x128, 0xf, 0xf, false);
    t = __builtin_amdgcn_permlane16_swap((unsigned)q, (unsigned)q, false, false);
    return (int)(t.x + t.y);
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
// prep: one WAVE per block of one shape group (four blocks per workgroup)
// ------------------------------------------------------------------------------------------------------------------
// event record: band | bitsAfter << 6 | nLines << 11 (nLines <= 2^20)
constexpr int kPrepWaves = 4;
__global__ __launch_bounds__(kWave * kPrepWaves) void chain_prep_kernel(
    DevShape S, int joint, int64_t nBlocks, const double* __restrict__ smr, const int* __restrict__ msSwitch,
    unsigned* __restrict__ evOut, unsigned* __restrict__ preOut,
    int forceFallback /* tests: scramble the candidate order first */) {
    __shared__ unsigned sEvAll[kPrepWaves][kMaxEvents];
    __shared__ unsigned sPreAll[kPrepWaves][kMaxEvents + 1];
    __shared__ double sKeyAll[kPrepWaves][kMaxEvents];
    __shared__ double sPhiAll[kPrepWaves][kWave], sSmrAll[kPrepWaves][kWave];
    __shared__ int sQAll[kPrepWaves][kWave], sSlotAll[kPrepWaves][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t blk = (int64_t)blockIdx.x * kPrepWaves + wave;
    if (blk >= nBlocks) return;                            // (wave-uniform; no workgroup barrier below)
    unsigned* sEv = sEvAll[wave];
    unsigned* sPre = sPreAll[wave];
    double* sKey = sKeyAll[wave];
    double* sPhi = sPhiAll[wave];
    double* sSmr = sSmrAll[wave];
    int* sQ = sQAll[wave];
    int* sSlot = sSlotAll[wave];
    const int nb = S.nBands;
    const int nsig = joint ? 4 : 1, nstream = joint ? 2 : 1, nTot = nstream * nb;
    const int K = S.maxMantBits - 1;
    const int nEv = nTot * K;
    // the sorted grant events.  lane i < nTot = (stream, band) i of bitalloc.py's concatenated arrays (codecThem.py:491-498);
    // stream 0 = Mid-or-Left, stream 1 = Side-or-Right per band (ms_stereo.py:70-81)
    const bool valid = lane < nTot;
    double s = 0.0;
    if (valid) {
        const int band = lane >= nb ? lane - nb : lane, strm = lane >= nb ? 1 : 0;
        const int sig = joint ? (msSwitch[blk * nb + band] ? 2 + strm : strm) : 0;
        s = smr[(blk * nsig + sig) * nb + band];
    }
    {
        double qd = floor(s / 6.0);
        qd = fmin(fmax(qd, -1000000.0), 1000000.0);         // (garbage in: still a bounded, valid candidate order)
        if (!(qd == qd)) qd = 0.0;
        double phi = s - 6.0 * qd;
        if (!(phi == phi)) phi = 0.0;
        sQ[lane] = (int)qd;
        sPhi[lane] = phi;
        sSmr[lane] = s;
        sSlot[lane] = 0;
    }
    wave_sync();
    // position of the band inside a 6 dB level: by SMR mod 6, larger first, equal ones by index
    int rank = 0;
    {
        const double myPhi = sPhi[lane];
        for (int j0 = 0; j0 < nTot; j0 += 8) {              // (eight broadcast reads in flight per trip)
            double pj[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pj[j] = sPhi[min(j0 + j, kWave - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                rank += (j0 + j < nTot && (pj[j] > myPhi || (pj[j] == myPhi && j0 + j < lane))) ? 1 : 0;
        }
    }
    if (valid) sSlot[min(rank, kWave - 1)] = lane;
    wave_sync();
    // from here on lane r holds the band with rank r
    const int band = valid ? sSlot[lane] : 0;
    const int nB = valid ? S.bandN[band >= nb ? band - nb : band] : 0;
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
            cur -= (k == 0) ? 12.0 : 6.0;
            ++k;
            nextLevel = k >= K ? INT_MIN : (k == 1 ? qB - 2 : nextLevel - 1);
        }
        base += __popcll(mask);
        costBase += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) sPre[nEv] = (unsigned)costBase;
    wave_sync();
    // the candidate order against the keys themselves (reads issued together: the arrays are padded to 15 per lane)
    bool bad = false;
    {
        double ka[kMaxEvents / kWave], kb[kMaxEvents / kWave];
        unsigned ea[kMaxEvents / kWave], eb[kMaxEvents / kWave];
#pragma unroll
        for (int j = 0; j < kMaxEvents / kWave; ++j) {
            const int p = min(lane + kWave * j, kMaxEvents - 2);
            ka[j] = sKey[p]; kb[j] = sKey[p + 1]; ea[j] = sEv[p]; eb[j] = sEv[p + 1];
        }
#pragma unroll
        for (int j = 0; j < kMaxEvents / kWave; ++j)
            bad |= (lane + kWave * j + 1 < nEv) && !event_before(ka[j], (int)(ea[j] & 63u), kb[j], (int)(eb[j] & 63u));
    }
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
                const int after = (int)((e >> 6) & 31u), nn = (int)(e >> 11);
                sPre[p] = run;
                run += (unsigned)(after == 2 ? 2 * nn : nn);
            }
            sPre[nEv] = run;
        }
        wave_sync();
    }
    unsigned* ev = evOut + blk * (int64_t)nEv;
    unsigned* pre = preOut + blk * (int64_t)(nEv + 1);
    for (int p = lane; p < nEv; p += kWave) ev[p] = sEv[p];
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
// What a thread holds of the coming items.  Stage 1 (one item ahead): the event list, the raw band peak, the raw lines of its
// units with the signal each comes from.  Stage 2 (two items ahead): the M/S switch of its band and an overall scale -- the
// lines of an item can only be requested once its M/S switch says which signal each band is coded from (ms_stereo.py:70-81,
// codecThem.py:524-551).
template <int NT> struct ItemRegs {
    unsigned ev[ChainDims<NT>::kEvPerThread], pre[ChainDims<NT>::kEvPerThread];
    double peak;
    double x[ChainDims<NT>::kUnitsPerThread][4];
    unsigned sigs[ChainDims<NT>::kUnitsPerThread];      // signal of each of the unit's four lines, 8 bits each
};
struct SwitchRegs { int sig, osc; };                    // thread i < nTot: M/S switch of its band (raw); thread q < 4: scale of signal q
// The fields of one group descriptor, held in scalar registers: they are read once per CHANGE of block shape (a stream is
// mostly runs of long blocks), not as dependent scalar loads inside every item.  The empty asm makes each value opaque,
// so the compiler keeps it in its register instead of loading it again from the descriptor where it is used.
// The pointers are declared in the GLOBAL address space: a pointer that comes out of memory is generic to the compiler, and
// loads through it would be flat_load instructions -- which count on the LDS counter as well, so that every wait for an LDS
// read would also wait for the prefetched loads still in flight (measured: +1.4 us per block).
#define MRC_GLOBAL __attribute__((address_space(1)))
typedef double double2n __attribute__((ext_vector_type(2)));       // (HIP's double2 / uint2 are classes: no address-space overloads)
typedef unsigned uint2n __attribute__((ext_vector_type(2)));
// Two views, so that both fit the scalar register file: what the COMPUTE of the current item reads, and what the LOADS of
// the next items read.
struct GroupView {                                      // compute side (current item)
    int joint, nb, nTot, M, K, nEv, maxN, nScaleBits, nstream;
    double budgetMono, budgetJointPre, blkswA, blkswB;
    MRC_GLOBAL int* bitAlloc;
    MRC_GLOBAL int* scaleFactor;
    MRC_GLOBAL unsigned short* mant;
    MRC_GLOBAL int* table;
};
struct LoadView {                                       // load side (next items)
    int joint, nb, nTot, M, nEv, nstream, nsig;
    const MRC_GLOBAL unsigned char* bandOfLine;
    const MRC_GLOBAL int* bandN;
    const MRC_GLOBAL double* lines;
    const MRC_GLOBAL double* peak;
    const MRC_GLOBAL int* oscale;
    const MRC_GLOBAL int* ms;
    const MRC_GLOBAL unsigned* ev;
    const MRC_GLOBAL unsigned* pre;
};
#define MRC_PIN(x) asm volatile("" : "+s"(x))
__device__ __forceinline__ GroupView group_view(const ChainGroupDev* __restrict__ groups, int g) {
    const ChainGroupDev& D = groups[g];
    GroupView V;
    V.joint = D.joint; V.nb = D.nb; V.nTot = D.nTot; V.M = D.M; V.K = D.K; V.nEv = D.nEv; V.maxN = D.maxN;
    V.nScaleBits = D.nScaleBits; V.nstream = D.nstream;
    V.budgetMono = D.budgetMono; V.budgetJointPre = D.budgetJointPre; V.blkswA = D.blkswA; V.blkswB = D.blkswB;
    V.bitAlloc = (MRC_GLOBAL int*)D.bitAlloc; V.scaleFactor = (MRC_GLOBAL int*)D.scaleFactor;
    V.mant = (MRC_GLOBAL unsigned short*)D.mant; V.table = (MRC_GLOBAL int*)D.table;
    MRC_PIN(V.joint); MRC_PIN(V.nb); MRC_PIN(V.nTot); MRC_PIN(V.M); MRC_PIN(V.K); MRC_PIN(V.nEv); MRC_PIN(V.maxN);
    MRC_PIN(V.nScaleBits); MRC_PIN(V.nstream);
    MRC_PIN(V.budgetMono); MRC_PIN(V.budgetJointPre); MRC_PIN(V.blkswA); MRC_PIN(V.blkswB);
    MRC_PIN(V.bitAlloc); MRC_PIN(V.scaleFactor); MRC_PIN(V.mant); MRC_PIN(V.table);
    return V;
}
__device__ __forceinline__ LoadView load_view(const ChainGroupDev* __restrict__ groups, int g) {
    const ChainGroupDev& D = groups[g];
    LoadView V;
    V.joint = D.joint; V.nb = D.nb; V.nTot = D.nTot; V.M = D.M; V.nEv = D.nEv; V.nstream = D.nstream; V.nsig = D.joint ? 4 : 1;
    V.bandOfLine = (const MRC_GLOBAL unsigned char*)D.bandOfLine; V.bandN = (const MRC_GLOBAL int*)D.bandN;
    V.lines = (const MRC_GLOBAL double*)D.lines; V.peak = (const MRC_GLOBAL double*)D.peak;
    V.oscale = (const MRC_GLOBAL int*)D.oscale; V.ms = (const MRC_GLOBAL int*)D.ms;
    V.ev = (const MRC_GLOBAL unsigned*)D.ev; V.pre = (const MRC_GLOBAL unsigned*)D.pre;
    MRC_PIN(V.joint); MRC_PIN(V.nb); MRC_PIN(V.nTot); MRC_PIN(V.M); MRC_PIN(V.nEv); MRC_PIN(V.nstream); MRC_PIN(V.nsig);
    MRC_PIN(V.bandOfLine); MRC_PIN(V.bandN); MRC_PIN(V.lines); MRC_PIN(V.peak); MRC_PIN(V.oscale); MRC_PIN(V.ms);
    MRC_PIN(V.ev); MRC_PIN(V.pre);
    return V;
}
// stage 2: the M/S switch of the thread's band and an overall scale, RAW -- what they mean (signal_of) is worked out when
// they are used, an iteration later: any arithmetic on the loaded values here would make the wave wait for them, and for
// every load of the prefetch in front of them, on the spot
__device__ __forceinline__ SwitchRegs switch_load(const LoadView& G, int64_t idx, int tid) {
    SwitchRegs W;
    W.sig = 0; W.osc = 0;
    if (tid < G.nTot && G.joint) W.sig = G.ms[idx * G.nb + (tid >= G.nb ? tid - G.nb : tid)];
    if (tid < G.nsig) W.osc = G.oscale[idx * G.nsig + tid];
    return W;
}
// the overall scale of signal sg, from the lanes q < 4 of the wave that hold the four scales (whole wave active)
__device__ __forceinline__ int osc_of(int osc, int sg) {
    const int o0 = __builtin_amdgcn_readlane(osc, 0), o1 = __builtin_amdgcn_readlane(osc, 1);
    const int o2 = __builtin_amdgcn_readlane(osc, 2), o3 = __builtin_amdgcn_readlane(osc, 3);
    return sg == 0 ? o0 : sg == 1 ? o1 : sg == 2 ? o2 : o3;
}
// ms_stereo.py:70-81: stream 0 carries Mid-or-Left, stream 1 Side-or-Right per band
__device__ __forceinline__ int signal_of(const LoadView& G, int tid, int msRaw) {
    return G.joint ? (msRaw ? 2 : 0) + (tid >= G.nb ? 1 : 0) : 0;
}
// the bands of the four lines of each of the thread's units (a function of the block shape only)
template <int NT>
__device__ __forceinline__ void unit_bands(const LoadView& G, int tid, unsigned* bands) {
    const int upl = G.M >> 2, nUnits = G.nstream * upl;
#pragma unroll
    for (int j = 0; j < ChainDims<NT>::kUnitsPerThread; ++j) {
        const int u = tid + NT * j;
        bands[j] = 0u;
        if (u < nUnits) bands[j] = *(const MRC_GLOBAL unsigned*)(G.bandOfLine + 4 * (u >= upl ? u - upl : u));
    }
}
// stage 1: everything else of the item; sSig = the item's signal table in LDS (written from its SwitchRegs)
template <int NT>
__device__ __forceinline__ void item_load(const LoadView& G, int64_t idx, int tid, const unsigned* bands,
                                          const int* __restrict__ sSig, ItemRegs<NT>& R) {
    constexpr int kEvPerThread = ChainDims<NT>::kEvPerThread, kUnitsPerThread = ChainDims<NT>::kUnitsPerThread;
    const int nEv = G.nEv, nTot = G.nTot, M = G.M, nb = G.nb;
    const MRC_GLOBAL unsigned* ev = G.ev + idx * (int64_t)nEv;
    const MRC_GLOBAL unsigned* pre = G.pre + idx * (int64_t)(nEv + 1);
#pragma unroll
    for (int j = 0; j < kEvPerThread; ++j) {
        const int p = tid + NT * j;
        R.ev[j] = p < nEv ? ev[p] : 0u;
        R.pre[j] = p <= nEv ? pre[p] : 0u;
    }
    R.peak = 0.0;
    if (tid < nTot) R.peak = G.peak[(idx * G.nsig + sSig[tid]) * nb + (tid >= nb ? tid - nb : tid)];
    const int upl = M >> 2, nUnits = G.nstream * upl;
    const MRC_GLOBAL double* blockLines = G.lines + idx * G.nsig * (int64_t)M;
#pragma unroll
    for (int j = 0; j < kUnitsPerThread; ++j) {
        const int u = tid + NT * j;
        R.sigs[j] = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) R.x[j][q] = 0.0;
        if (u < nUnits) {
            const int strm = u >= upl ? 1 : 0;
            const int k = 4 * (u - strm * upl);
            unsigned sg = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) sg |= (unsigned)sSig[strm * nb + (int)((bands[j] >> (8 * q)) & 0xffu)] << (8 * q);
            R.sigs[j] = sg;
            if (sg == (sg & 0xffu) * 0x01010101u) {        // one signal for the four lines (nearly always): two 16-byte loads
                const MRC_GLOBAL double* src = blockLines + (int64_t)(sg & 0xffu) * M + k;
                const double2n a = *(const MRC_GLOBAL double2n*)src, b = *(const MRC_GLOBAL double2n*)(src + 2);
                R.x[j][0] = a.x; R.x[j][1] = a.y; R.x[j][2] = b.x; R.x[j][3] = b.y;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) R.x[j][q] = blockLines[(int64_t)((sg >> (8 * q)) & 0xffu) * M + k + q];
            }
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
    __shared__ int sSig[2][kWave];                       // signal of (stream, band) i: this item's / the next item's (by parity)
    __shared__ int sOsc[2][4];                           // overall scale of signal q, likewise
    __shared__ unsigned sInfo[kWave];                    // per (stream, band): bits | scale factor << 8
    __shared__ unsigned sEsc[kWave];                     // per (stream, band): bits + escape code length of each table, 8 bits each
    __shared__ unsigned sLut[kLutSize + 1];              // per value: the four code lengths, 8 bits each (0: not in the table)
    __shared__ int sCtl[4];                              // remaining bits, raw bits of stream 0 / 1
    __shared__ int sCut;                                 // events that are certain grants (counted by all waves)
    __shared__ unsigned sAcc[2][4];                      // Huffman prices of the block, summed over the waves (LDS atomics); by item parity
    __shared__ int sItems[256];                          // ring of item ids (see below)
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
    // the reservoir travels in a register of EVERY thread: each wave works the table decision out for itself from the block's
    // price sums, so that no barrier stands between the decision and the next item's budget
    int resReg = reservoir[strmId];
    if (tid == 0) sCut = 0;
    if (tid < 8) sAcc[tid >> 2][tid & 3] = 0u;
    if (tid < kWave) sBits[tid] = 0;
    const long long i0 = itemStart[strmId], i1 = itemStart[strmId + 1];
    ItemRegs<NT> R;
    SwitchRegs W;                                        // the NEXT item's switch and scales
    W.sig = 0; W.osc = 0;
    unsigned bandsCur[kUnitsPerThread], bandsNext[kUnitsPerThread];   // bands of the thread's lines: current / next block shape
    int nCur = 0, nNext = 0;                             // lines of the thread's (stream, band)
    int itemCur = 0, itemNext = 0, itemAfter = 0;
    // wave 0, lane = (stream, band): the overall scale of the band's signal, this item's / the next item's -- picked from
    // the lanes that hold the four scales when the switch arrives, so that the scale factors need no LDS round trips
    int oscCur = 0, oscNext = 0;
    GroupView G = group_view(groups, 0);                 // the current item's group (compute side)
    LoadView Ln = load_view(groups, 0);                  // the next item's group (load side)
    int gOfG = 0, gOfLn = 0;
    auto band_size = [&](const LoadView& V) { return tid < V.nTot ? (int)V.bandN[tid >= V.nb ? tid - V.nb : tid] : 0; };
    // The stream's item ids sit in an LDS ring of two halves of 128, refilled half by half (once per 128 items, by a
    // load whose result is used inside the refill branch only).  A scalar load in the loop would share its counter with the
    // LDS reads -- every LDS wait would also wait for it -- and a vector load whose result is picked up iterations later
    // makes the compiler wait for ALL outstanding vector memory operations, the previous item's stores included, every
    // iteration.
    constexpr int kIdHalf = 128;
    auto fill_ids = [&](long long chunk) {                // ids [i0 + 128 chunk, + 128) -> half (chunk & 1)
        if (tid < kIdHalf) {
            const long long i = i0 + chunk * kIdHalf + tid;
            sItems[(int)(chunk & 1) * kIdHalf + tid] = i < i1 ? items[i] : 0;
        }
    };
    fill_ids(0);
    fill_ids(1);
    __syncthreads();
    auto item_id = [&](long long i) {                     // (uniform) i inside the two chunks held
        return __builtin_amdgcn_readfirstlane(sItems[(int)((i - i0) & (2 * kIdHalf - 1))]);
    };
    if (i0 < i1) {
        itemCur = item_id(i0);
        itemNext = i0 + 1 < i1 ? item_id(i0 + 1) : 0;
        itemAfter = i0 + 2 < i1 ? item_id(i0 + 2) : 0;
        gOfG = (int)((unsigned)itemCur >> 28);
        if (gOfG != 0) { G = group_view(groups, gOfG); Ln = load_view(groups, gOfG); }
        gOfLn = gOfG;
        const SwitchRegs W0 = switch_load(Ln, itemCur & 0x0fffffff, tid);
        if (tid < kWave) { sSig[0][tid] = signal_of(Ln, tid, W0.sig); oscCur = osc_of(W0.osc, signal_of(Ln, tid, W0.sig)); }
        if (tid < 4) sOsc[0][tid] = W0.osc;
        unit_bands<NT>(Ln, tid, bandsCur);
        nCur = band_size(Ln);
        __syncthreads();
        item_load<NT>(Ln, itemCur & 0x0fffffff, tid, bandsCur, sSig[0], R);
#pragma unroll
        for (int j = 0; j < kUnitsPerThread; ++j) bandsNext[j] = bandsCur[j];
        nNext = nCur;
        if (i0 + 1 < i1) {
            const int gn = (int)((unsigned)itemNext >> 28);
            if (gn != gOfLn) { Ln = load_view(groups, gn); gOfLn = gn; unit_bands<NT>(Ln, tid, bandsNext); nNext = band_size(Ln); }
            W = switch_load(Ln, itemNext & 0x0fffffff, tid);
        }
    }
    __syncthreads();
#ifdef MRC_CHAIN_PROFILE
    long long tProf_ = clock64();
#endif
    for (long long it = i0; it < i1; ++it) {
        const int item = itemCur;
        const int64_t idx = item & 0x0fffffff;
        const int nb = G.nb, nTot = G.nTot, M = G.M, nEv = G.nEv, nstream = G.nstream;
        MRC_CP(0);
        // ---- this item's event list and peaks: registers -> LDS (its lines stay in registers); the next item's switch
        //      and scales: registers -> the other half of the signal tables
        const int par = (int)((it - i0) & 1);
#pragma unroll
        for (int j = 0; j < kEvPerThread; ++j) {
            const int p = tid + kChainThreads * j;
            sEv[p] = R.ev[j]; sPre[p] = R.pre[j];
        }
        const double peakCur = R.peak;                   // (wave 0: raw max |X| of the lane's (stream, band))
        if (tid < kWave) {
            const int sgN = signal_of(Ln, tid, W.sig);
            sSig[par ^ 1][tid] = sgN;
            oscNext = osc_of(W.osc, sgN);
        }
        if (tid < 4) sOsc[par ^ 1][tid] = W.osc;
        // ---- bit allocation (bitalloc.py:106-155), its HEAD by all waves, straight from the registers that hold the event
        //      list: the budget of codecThem.py:299-308 / 381-396 from the reservoir the previous item left; the events whose
        //      cost prefix leaves at least max(nLines) bits are certain grants -- counted (the cut), and every band's bits
        //      raised to what its last such grant gives.  (nLines <= left <=> nLines + spent <= Bf; left > 0 <=> spent < Bc)
        double budget;
        {
            const double r = (double)resReg;
            if (G.joint) { budget = G.budgetJointPre + r; budget -= G.blkswA; budget -= G.blkswB; }
            else budget = G.budgetMono + r;
        }
        const int Bf = (int)fmin(fmax(floor(budget), -1.0e9), 1.0e9), Bc = (int)fmin(fmax(ceil(budget), -1.0e9), 1.0e9);
        {
            const int maxN = G.maxN;
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < kEvPerThread; ++j) {
                const int p = tid + kChainThreads * j;
                if (p < nEv && (int)R.pre[j] + maxN <= Bf) {
                    ++cnt;
                    atomicMax(&sBits[R.ev[j] & 63u], (int)((R.ev[j] >> 6) & 31u));
                }
            }
            cnt = wave_sum_i(cnt);
            if (lane == 0 && cnt) atomicAdd(&sCut, cnt);
        }
        double xr[kUnitsPerThread][4];
        unsigned sigsOf[kUnitsPerThread];
#pragma unroll
        for (int j = 0; j < kUnitsPerThread; ++j) {
            sigsOf[j] = R.sigs[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[j][q] = R.x[j][q];
        }
        MRC_CP(1);
        __syncthreads();
        MRC_CP(2);
        if (tid < 4) sAcc[par ^ 1][tid] = 0u;            // the next item's sums (last read in the previous item's decision)
        // ---- the next item's loads go out now and land while this one is computed; and the switch of the item after it
        // (ids up to it + 3 are read here; a chunk is replaced when the scan is 8 items into the NEXT one: everything of the
        // old chunk has been read, and the new ids are visible after this iteration's barriers, 117 items before their turn)
        if (((it - i0) & (kIdHalf - 1)) == 8 && it - i0 >= kIdHalf) fill_ids((it - i0) / kIdHalf + 1);
        const int itemAfter2 = it + 3 < i1 ? item_id(it + 3) : 0;
        if (it + 1 < i1) item_load<NT>(Ln, itemNext & 0x0fffffff, tid, bandsNext, sSig[par ^ 1], R);
        const int gAfter = (int)((unsigned)itemAfter >> 28);
        if (it + 2 < i1) {
            if (gAfter == gOfLn) {
                W = switch_load(Ln, itemAfter & 0x0fffffff, tid);
            } else {                                                            // (a change of block shape: rare)
                W = switch_load(load_view(groups, gAfter), itemAfter & 0x0fffffff, tid);
                // the values are waited for HERE: a load still pending at the end of a rare path would make the compiler
                // guard the code behind the merge with a wait for all vector memory operations, on every iteration
                asm volatile("" : "+v"(W.sig), "+v"(W.osc));
            }
        }
        MRC_CP(3);
        if (wave == 0) {
            // ---- the rest of the allocation: the tail behind the cut
            const int cut = sCut;
            const bool valid = lane < nTot;
            int myBits = valid ? sBits[lane] : 0;
            const int myN = valid ? nCur : 0;
            int spent = (int)sPre[cut];
            MRC_CP(4);
            // The tail, in batches: the next 64 events at a time.  A band that no longer fits can never be granted again
            // (the bits left only shrink), so it is retired at once -- as bitalloc.py:149-151 would retire it when it next
            // came up -- and every event of a live band is a grant UNLESS the grants of live events before it in the batch
            // have used its room up: a prefix sum over the batch's live costs finds the first such event; everything before
            // it is granted in one go, the live set is brought up to date, and the rest of the batch is looked at again.
            unsigned long long alive = nTot >= 64 ? ~0ull : ((1ull << nTot) - 1ull);
            alive &= ~__ballot(valid && myN + spent > Bf);
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
            sBits[lane] = 0;                                                   // (for the next item's head)
            if (lane == 0) sCut = 0;
            MRC_CP(5);
#ifdef MRC_CHAIN_PROFILE
            if (tid == 0) { atomicAdd(&gChainProf[12], (unsigned long long)(e - cut)); atomicAdd(&gChainProf[13], 1ull); }
#endif
            // ---- scale factors (codecThem.py:346-347), raw size of each stream (codecThem.py:141-146)
            const int rawMine = myBits * myN;
            const int raw01 = wave_sum2_i((valid && lane < nb) ? rawMine : 0, (valid && lane >= nb) ? rawMine : 0);
            if (valid) {
                // codecThem.py:346-347: the scale factor comes from max |scaled line| of the band; scaling by 2^overallScale
                // is exact, so maximum and scaling commute
                const int sf = scale_factor32(ldexp(peakCur, oscCur), G.nScaleBits, myBits);
                sInfo[lane] = (unsigned)myBits | ((unsigned)sf << 8);
                sEsc[lane] = escLen4 + (unsigned)myBits * 0x01010101u;
                G.bitAlloc[idx * nTot + lane] = myBits;
                G.scaleFactor[idx * nTot + lane] = sf;
            }
            if (lane == 0) {
                sCtl[0] = (int)(budget - (double)spent);                   // int(bitsLeft): truncation toward zero (bitalloc.py:155)
                sCtl[1] = raw01;                                           // (lanes 0-31: stream 0)
            }
            if (lane == 32) sCtl[2] = raw01;                               // (lanes 32-63: stream 1)
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
                    const unsigned bands = bandsCur[jj];
                    double x[4];                                           // codecThem.py:323: X * 2^overallScale (exact)
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = ldexp(xr[jj][q], sOsc[par][(sigsOf[jj] >> (8 * q)) & 0xffu]);
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
                    uint2n w;
                    w.x = code[0] | (code[1] << 16);
                    w.y = code[2] | (code[3] << 16);
                    *(MRC_GLOBAL uint2n*)(G.mant + (idx * nstream + strm) * (int64_t)M + k) = w;
                }
            }
        }
        {
            // bytes -> 16-bit fields (per table at most 25 bits x 1024 lines): tables 0 | 2 and 1 | 3 of each stream
            const unsigned w = wave_sum4_u(accA & 0x00ff00ffu, (accA >> 8) & 0x00ff00ffu, accB & 0x00ff00ffu, (accB >> 8) & 0x00ff00ffu);
            if ((lane & 15) == 0) atomicAdd(&sAcc[par][lane >> 4], w);         // row r of the wave: field pair r
        }
        MRC_CP(8);
        __syncthreads();
        MRC_CP(9);
        {
            // every wave for itself (uniform values); thread 0 writes the tables down
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = sAcc[par][q];
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
                    if (tid == 0) G.table[idx * nstream + s] = table;
                    res += raw - best;                                    // codecThem.py:202,224,274
                }
            }
            resReg = res;
            if (tid == 0 && resTrace) resTrace[it] = res;
        }
        oscCur = oscNext;
        itemCur = itemNext;
        itemNext = itemAfter;
        itemAfter = itemAfter2;
#pragma unroll
        for (int j = 0; j < kUnitsPerThread; ++j) bandsCur[j] = bandsNext[j];
        nCur = nNext;
        if (it + 1 < i1 && gOfLn != gOfG) { G = group_view(groups, gOfLn); gOfG = gOfLn; }     // (changes of block shape: rare)
        if (it + 2 < i1 && gAfter != gOfLn) {
            Ln = load_view(groups, gAfter); gOfLn = gAfter;
            unit_bands<NT>(Ln, tid, bandsNext);
            nNext = band_size(Ln);
#pragma unroll
            for (int j = 0; j < kUnitsPerThread; ++j) asm volatile("" : "+v"(bandsNext[j]));     // (waited for here, see above)
            asm volatile("" : "+v"(nNext));
        }
        MRC_CP(10);
        // (no barrier here: what the next item writes before its first barrier -- the event list, peaks, the other half of the
        // signal tables, the head's counters -- was last read before this item's barrier 2 or 3; the price sums alternate)
        MRC_CP(11);
    }
    if (tid == 0) reservoir[strmId] = resReg;
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
// (hdrLen == 0: no headers), and where every stream's bytes start: what the host wants to know without reading the
// positions of all chunks back
__global__ void chain_header_kernel(int64_t nStreams, int hdrLen, const unsigned char* __restrict__ hdr,
                                    const long long* __restrict__ firstChunk, const long long* __restrict__ pos,
                                    unsigned char* __restrict__ out, long long outCap, long long* __restrict__ streamPos) {
    const int64_t s = blockIdx.x;
    const long long p0 = pos[firstChunk[s]] - hdrLen;
    if (threadIdx.x == 0) streamPos[s] = p0;
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

hipError_t launch_chain_prep(const DevShape& S, int joint, int64_t nBlocks, const double* smr, const int* msSwitch,
                             unsigned* ev, unsigned* pre, int forceFallback, hipStream_t st) {
    if (nBlocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_prep_kernel, dim3((unsigned)((nBlocks + kPrepWaves - 1) / kPrepWaves)), dim3(kWave * kPrepWaves),
                       0, st, S, joint, nBlocks, smr, msSwitch, ev, pre, forceFallback);
    return hipGetLastError();
}

hipError_t launch_chain_phase_b(int64_t nStreams, const ChainGroupDev* groups, const int* items, const long long* itemStart,
                                int* reservoir, int* resTrace, int useHuffman, int threads, hipStream_t st) {
    if (nStreams <= 0) return hipSuccess;
    // a workgroup per stream.  Few streams: large workgroups (the chip is idle anyway, the stream's latency is what
    // counts); many: small ones, eight streams per CU
    if (threads <= 0) threads = nStreams <= 512 ? 512 : 256;   // (measured on one stream: 3.7 / 3.3 / 4.3 us per block at 256 / 512 / 1024)
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
                                const long long* pos, unsigned char* out, long long outCap, long long* streamPos,
                                hipStream_t st) {
    if (nStreams <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_header_kernel, dim3((unsigned)nStreams), dim3(64), 0, st, nStreams, hdrLen, hdr, firstChunk,
                       pos, out, outCap, streamPos);
    return hipGetLastError();
}

}  // namespace mrc
