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
            sPos[band * K + k] = (unsigned short)p;
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
                sPos[bnd * K + after - 2] = (unsigned short)p;
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
__global__ __launch_bounds__(kChainThreads) void chain_phase_b_kernel(
    const ChainGroupDev* __restrict__ groups, const int* __restrict__ items, const long long* __restrict__ itemStart,
    int* __restrict__ reservoir, int* __restrict__ resTrace /* nullable: reservoir after every item */, int useHuffman) {
    __shared__ unsigned sEv[kMaxEvents];
    __shared__ unsigned sPre[kMaxEvents + 1];
    __shared__ unsigned short sPos[kMaxEvents];
    __shared__ double sPeak[kWave];
    __shared__ unsigned sInfo[kWave];                    // per (stream, band): bits | scale factor << 8
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
    if (tid == 0) sCtl[3] = reservoir[strmId];
    __syncthreads();
    const long long i0 = itemStart[strmId], i1 = itemStart[strmId + 1];
    for (long long it = i0; it < i1; ++it) {
        const int item = items[it];
        const ChainGroupDev& G = groups[(unsigned)item >> 28];
        const int64_t idx = item & 0x0fffffff;
        const int nb = G.nb, nTot = G.nTot, M = G.M, K = G.K, nEv = G.nEv, nstream = G.nstream;
        // ---- stage the block's event list and peaks
        {
            const unsigned* ev = G.ev + idx * (int64_t)nEv;
            const unsigned* pre = G.pre + idx * (int64_t)(nEv + 1);
            const unsigned short* pos = G.pos + idx * (int64_t)nEv;
            for (int p = tid; p < nEv; p += kChainThreads) { sEv[p] = ev[p]; sPos[p] = pos[p]; }
            for (int p = tid; p <= nEv; p += kChainThreads) sPre[p] = pre[p];
            if (tid < nTot) sPeak[tid] = G.peakSel[idx * nTot + tid];
        }
        __syncthreads();
        if (wave == 0) {
            // ---- bit allocation (bitalloc.py:106-155) for the budget of codecThem.py:299-308 / 381-396
            const double r = (double)sCtl[3];
            double budget;
            if (G.joint) { budget = G.budgetJointPre + r; budget -= G.blkswA; budget -= G.blkswB; }
            else budget = G.budgetMono + r;
            const double bfD = fmin(fmax(floor(budget), -2.0e9), 2.0e9), bcD = fmin(fmax(ceil(budget), -2.0e9), 2.0e9);
            const int Bf = (int)bfD, Bc = (int)bcD;      // nLines <= left  <=>  nLines + spent <= Bf;  left > 0  <=>  spent < Bc
            // how many events are certain grants: fewer than maxN bits have been spent short of the budget
            int cnt = 0;
            for (int p = lane; p < nEv; p += kWave) cnt += ((long long)sPre[p] + G.maxN <= (long long)Bf) ? 1 : 0;
            const int cut = wave_sum_i(cnt);
            const bool valid = lane < nTot;
            int myBits = 0;
            if (valid) {
                int c = 0;
                for (int k = 0; k < K; ++k) c += (int)sPos[lane * K + k] < cut ? 1 : 0;
                myBits = c ? c + 1 : 0;                  // the first grant gives two bits
            }
            const int myN = valid ? G.bandN[lane % nb] : 0;
            long long spent = sPre[cut];
            // the tail, event by event (uniform control flow: the event, the budget and the retired set live in scalars)
            unsigned long long alive = nTot >= 64 ? ~0ull : ((1ull << nTot) - 1ull);
            alive &= ~__ballot(valid && (long long)myN + spent > (long long)Bf);   // can never be granted again: retired when they come up
            int e = cut;
            bool done = !(spent < (long long)Bc) || alive == 0ull;
            while (!done && e < nEv) {
                const unsigned rec = (e + lane < nEv) ? sEv[e + lane] : 0u;
                const int jmax = min(kWave, nEv - e);
                for (int j = 0; j < jmax && !done; ++j) {
                    const unsigned rj = (unsigned)__builtin_amdgcn_readlane((int)rec, j);
                    const int bnd = (int)(rj & 63u);
                    if (!((alive >> bnd) & 1ull)) continue;
                    const int nn = (int)(rj >> 11), after = (int)((rj >> 6) & 31u);
                    if ((long long)nn + spent <= (long long)Bf) {          // bitalloc.py:134 (only nLines is tested, also for the 2-bit grant)
                        spent += after == 2 ? 2 * nn : nn;
                        myBits = lane == bnd ? after : myBits;
                        alive &= ~__ballot(valid && (long long)myN + spent > (long long)Bf);
                    } else {
                        alive &= ~(1ull << bnd);                           // bitalloc.py:149-151
                    }
                    done = !(spent < (long long)Bc) || alive == 0ull;
                }
                e += kWave;
            }
            // ---- scale factors (codecThem.py:346-347), raw size of each stream (codecThem.py:141-146)
            const int rawMine = myBits * myN;
            const int raw0 = wave_sum_i((valid && lane < nb) ? rawMine : 0);
            const int raw1 = wave_sum_i((valid && lane >= nb) ? rawMine : 0);
            if (valid) {
                const int sf = scale_factor_dev(sPeak[lane], G.nScaleBits, myBits);
                sInfo[lane] = (unsigned)myBits | ((unsigned)sf << 8);
                G.bitAlloc[idx * nTot + lane] = myBits;
                G.scaleFactor[idx * nTot + lane] = sf;
            }
            if (lane == 0) {
                sCtl[0] = (int)(budget - (double)spent);                   // int(bitsLeft): truncation toward zero (bitalloc.py:155)
                sCtl[1] = raw0;
                sCtl[2] = raw1;
            }
        }
        __syncthreads();
        // ---- mantissas (codecThem.py:348-349) and the price of every Huffman table (codecThem.py:157-173)
        int cA[4] = {0, 0, 0, 0}, cB[4] = {0, 0, 0, 0};
        {
            const int upl = M >> 2;                                        // units of four lines per stream
            const int nUnits = nstream * upl;
            for (int u = tid; u < nUnits; u += kChainThreads) {
                const int strm = u >= upl ? 1 : 0;
                const int k = 4 * (u - strm * upl);
                const unsigned bands = *reinterpret_cast<const unsigned*>(G.bandOfLine + k);
                const double* src = G.xsel + (idx * nstream + strm) * (int64_t)M + k;
                const double2 p = *reinterpret_cast<const double2*>(src), q = *reinterpret_cast<const double2*>(src + 2);
                const double x[4] = {p.x, p.y, q.x, q.y};
                unsigned short code[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned info = sInfo[strm * nb + ((bands >> (8 * j)) & 0xffu)];
                    const int ba = (int)(info & 0xffu);
                    int c = 0;
                    if (ba) {
                        c = mantissa_dev(x[j], (int)(info >> 8), G.nScaleBits, ba);
                        const unsigned lens = sLut[c < kLutSize ? c : kLutSize];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int len = (int)((lens >> (8 * t)) & 0xffu);
                            // a value without a code costs the escape code + the raw mantissa; the escape VALUE itself is
                            // priced as its code alone (codecThem.py:169-172)
                            const int add = len ? len : ba + (int)kChainCodeLen[t][kChainEscape[t]];
                            if (strm) cB[t] += add; else cA[t] += add;
                        }
                    }
                    code[j] = (unsigned short)c;
                }
                uint2 w;
                w.x = (unsigned)code[0] | ((unsigned)code[1] << 16);
                w.y = (unsigned)code[2] | ((unsigned)code[3] << 16);
                *reinterpret_cast<uint2*>(G.mant + (idx * nstream + strm) * (int64_t)M + k) = w;
            }
        }
        {
            // per table at most 25 bits x 1024 lines: two 16-bit fields per word
            const unsigned w0 = (unsigned)wave_sum_i(cA[0] | (cA[1] << 16)), w1 = (unsigned)wave_sum_i(cA[2] | (cA[3] << 16));
            const unsigned w2 = (unsigned)wave_sum_i(cB[0] | (cB[1] << 16)), w3 = (unsigned)wave_sum_i(cB[2] | (cB[3] << 16));
            if (lane == 0) { sRed[wave][0] = w0; sRed[wave][1] = w1; sRed[wave][2] = w2; sRed[wave][3] = w3; }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned w[4] = {0, 0, 0, 0};
            for (int v = 0; v < kChainThreads / kWave; ++v)
                for (int q = 0; q < 4; ++q) w[q] += sRed[v][q];
            int res = sCtl[0];
            for (int s = 0; s < nstream; ++s) {
                const int raw = sCtl[1 + s];
                const int cost[4] = {(int)(w[2 * s] & 0xffffu), (int)(w[2 * s] >> 16), (int)(w[2 * s + 1] & 0xffffu),
                                     (int)(w[2 * s + 1] >> 16)};
                int best = raw, table = 15;                               // codecThem.py:147-149
                if (useHuffman)
                    for (int t = 0; t < 4; ++t)
                        if (cost[t] < best) { best = cost[t]; table = t; }  // strictly less: raw, then the first table, win ties
                G.table[idx * nstream + s] = table;
                res += raw - best;                                        // codecThem.py:202,224,274
            }
            sCtl[3] = res;
            if (resTrace) resTrace[it] = res;
        }
        __syncthreads();
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
                                int* reservoir, int* resTrace, int useHuffman, hipStream_t st) {
    if (nStreams <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_phase_b_kernel, dim3((unsigned)nStreams), dim3(kChainThreads), 0, st, groups, items,
                       itemStart, reservoir, resTrace, useHuffman);
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
