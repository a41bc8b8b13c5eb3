// Back end of the encode path on gfx950, after the lines and SMRs exist:
//   band_stats_kernel   the per-band max |X| that the scale factors need (codecThem.py:346) for the stage entry points
//                       (the full path gets them from smr_kernel); the M/S decision (ms_stereo.py:5-27) is
//                       ms_switch_kernel in mrc_kernels.hip
//   bitalloc_kernel     ms_stereo.py:70-81 (SMR select) + budgets (codecThem.py:299-308, 381-396) +
//                       bitalloc.py:106-155 -- ONE LANE PER FRAME: the greedy loop is serial per frame, so 64
//                       frames share a wavefront and each lane scans its own <= 64 running SMRs in LDS
//   quantize_kernel     quantize.py:114-146 (per-band scale factor with nMantBits = allocation) and
//                       quantize.py:294-322 (mantissas), codecThem.py:335-350 / 510-559 -- one wavefront per
//                       (frame, stream), 16 lines per lane, coalesced int32 stores
// Integer-deciding float64 arithmetic keeps the reference's operation order (file built with
// -ffp-contract=off).
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
namespace {

constexpr int kBandLds = 2048;                          // lines per block whose band map quantize_kernel stages in LDS

// ------------------------------------------------------------------------------------------------
// band statistics: one wavefront per frame
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void band_stats_kernel(DevShape S, int joint, int wantPeaks,
                                                           const double* __restrict__ lines,
                                                           int* __restrict__ msSwitch, double* __restrict__ bandPeak) {
    __shared__ unsigned long long peakBits[4 * kMaxBands];
    const int lane = threadIdx.x;
    const int64_t f = blockIdx.x;
    const int M = S.halfN, nb = S.nBands;
    const int nsig = joint ? 4 : 1;
    const double* X = lines + f * nsig * M;
    for (int i = lane; i < nsig * nb; i += kWave) peakBits[i] = 0ull;
    __syncthreads();
    // max |X| per band: |x| >= 0, so the raw bit pattern orders like the value
    for (int s = 0; s < (wantPeaks ? nsig : 0); ++s)
        for (int k = lane; k < M; k += kWave)
            atomicMax(&peakBits[s * nb + S.bandOfLine[k]],
                      (unsigned long long)__double_as_longlong(fabs(X[s * M + k])));
    __syncthreads();
    for (int i = lane; i < (wantPeaks ? nsig * nb : 0); i += kWave)
        bandPeak[f * nsig * nb + i] = __longlong_as_double((long long)peakBits[i]);
}

// ------------------------------------------------------------------------------------------------
// bit allocation (bitalloc.py:106-155), one lane per problem.  run[] / bits[] are this wave's LDS slabs laid out
// [band][lane]: a lane's scan over bands is conflict-free and so is its indexed update.
// np.argmax over all bands every iteration is what the loop costs (it is bound by instruction issue), so the maximum is
// kept in TWO LEVELS: bands are cut into groups of gs consecutive bands whose first maximum (value, index) sits in the
// slabs gv / gi; an iteration compares the nG group maxima, updates one band and rescans only that band's group.
// "First maximum wins" survives: strict > inside a group in band order, strict > across groups in group order.
// ------------------------------------------------------------------------------------------------
struct AllocSlabs {
    double* run;            // [nTot][ld]
    double* gv;             // [nG][ld]
    unsigned char* bits;    // [nTot][ld]
    unsigned char* gi;      // [nG][ld]
    int* nLines;            // [nTot]
    int gs, nG;
};
__host__ __device__ constexpr int alloc_group_size(int nTot) { return nTot <= 16 ? 4 : nTot <= 32 ? 6 : 8; }
__host__ __device__ inline size_t alloc_lds_bytes(int nTot, int ld) {
    const int gs = alloc_group_size(nTot), nG = (nTot + gs - 1) / gs;
    return (size_t)(nTot + nG) * ld * sizeof(double) + (((size_t)(nTot + nG) * ld + 3) & ~(size_t)3) + (size_t)nTot * sizeof(int);
}
__device__ __forceinline__ AllocSlabs alloc_slabs(double* lds, int nTot, int ld) {
    AllocSlabs a;
    a.gs = alloc_group_size(nTot);
    a.nG = (nTot + a.gs - 1) / a.gs;
    a.run = lds;
    a.gv = lds + nTot * ld;
    a.bits = reinterpret_cast<unsigned char*>(lds + (nTot + a.nG) * ld);
    a.gi = a.bits + nTot * ld;
    a.nLines = reinterpret_cast<int*>(a.bits + (((nTot + a.nG) * ld + 3) & ~3));
    return a;
}

// NTOT > 0: the number of bands (x streams) as a compile-time constant -- group size, group count and the row stride (64
// lanes) fold, the scans over a group and over the group maxima unroll (their LDS reads go out together instead of one
// per loop trip); 0: taken from the arguments.
template <int NTOT>
__device__ __forceinline__ void alloc_rescan(const AllocSlabs& A, int G, int nTotArg, int lane, int ldArg) {
    const int nTot = NTOT ? NTOT : nTotArg, ld = NTOT ? kWave : ldArg;
    const int gs = NTOT ? alloc_group_size(NTOT) : A.gs;
    const int b0 = G * gs;
    double v0 = A.run[b0 * ld + lane];
    int i0 = b0;
    if (NTOT) {
        double v[8];
#pragma unroll
        for (int j = 1; j < (NTOT ? alloc_group_size(NTOT ? NTOT : 1) : 1); ++j) v[j] = A.run[min(b0 + j, nTot - 1) * ld + lane];
#pragma unroll
        for (int j = 1; j < (NTOT ? alloc_group_size(NTOT ? NTOT : 1) : 1); ++j)
            if (b0 + j < nTot && v[j] > v0) { v0 = v[j]; i0 = b0 + j; }
    } else {
        const int b1 = min(b0 + gs, nTot);
        for (int b = b0 + 1; b < b1; ++b) {
            const double v = A.run[b * ld + lane];
            if (v > v0) { v0 = v; i0 = b; }
        }
    }
    A.gv[G * ld + lane] = v0;
    A.gi[G * ld + lane] = (unsigned char)i0;
}

template <int NTOT = 0>
__device__ __forceinline__ double bitalloc_lane(const AllocSlabs& A, int nTotArg, int maxMantBits, double budget, int lane,
                                                bool active, int ldArg = kWave) {
    const int nTot = NTOT ? NTOT : nTotArg, ld = NTOT ? kWave : ldArg;
    constexpr int kG = NTOT ? (NTOT + alloc_group_size(NTOT ? NTOT : 1) - 1) / alloc_group_size(NTOT ? NTOT : 1) : 0;
    const int nG = NTOT ? kG : A.nG;
    double left = budget;
    int retired = 0;
    // every iteration grants (<= maxMantBits-1 times per band) or retires (<= nTot times): the loop ends by
    // itself; the counter only guards against spinning on NaN input
    int guard = (maxMantBits + 2) * nTot + 8;
    bool live = active && left > 0;
    if (live)
        for (int G = 0; G < nG; ++G) alloc_rescan<NTOT>(A, G, nTot, lane, ld);
    while (__any(live)) {
        if (live) {
            double best = A.gv[lane];
            int G = 0;
            if (NTOT) {
                double gvv[kG ? kG : 1];
#pragma unroll
                for (int g = 1; g < kG; ++g) gvv[g] = A.gv[g * ld + lane];
#pragma unroll
                for (int g = 1; g < kG; ++g)
                    if (gvv[g] > best) { best = gvv[g]; G = g; }        // np.argmax: first maximum wins
            } else {
                for (int g = 1; g < nG; ++g) {
                    const double v = A.gv[g * ld + lane];
                    if (v > best) { best = v; G = g; }                  // np.argmax: first maximum wins
                }
            }
            const int idx = A.gi[G * ld + lane];
            const int have = A.bits[idx * ld + lane];
            const int n = A.nLines[idx];
            if (have < maxMantBits && (double)n <= left) {
                if (have == 0) {
                    A.bits[idx * ld + lane] = 2;
                    left -= (double)(2 * n);
                    A.run[idx * ld + lane] = best - 12.0;
                } else {
                    A.bits[idx * ld + lane] = (unsigned char)(have + 1);
                    left -= (double)n;
                    A.run[idx * ld + lane] = best - 6.0;
                }
            } else {
                A.run[idx * ld + lane] = -99999999999999999.0;
                if (++retired == nTot) live = false;
            }
            if (!(left > 0) || --guard <= 0) live = false;
            if (live) alloc_rescan<NTOT>(A, G, nTot, lane, ld);
        }
    }
    return left;
}

// fpw = frames per wave (<= 64; the launcher uses 64, see there).
template <int NTOT>
__global__ __launch_bounds__(kWave) void bitalloc_kernel(DevShape S, int joint, int64_t nFrames, int fpw,
                                                         const double* __restrict__ smr,
                                                         const int* __restrict__ msSwitch,
                                                         const int* __restrict__ resIn, int* __restrict__ bitAlloc,
                                                         int* __restrict__ resOut) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int nb = S.nBands, nsig = joint ? 4 : 1, nstream = joint ? 2 : 1;
    const int nTot = nstream * nb;
    const AllocSlabs A = alloc_slabs(lds, nTot, fpw);
    const int64_t f = (int64_t)blockIdx.x * fpw + lane;
    const bool active = lane < fpw && f < nFrames;
    for (int i = lane; i < nTot; i += kWave) A.nLines[i] = S.bandN[i % nb];
    // stream 0 = Mid-or-Left, stream 1 = Side-or-Right (ms_stereo.py:70-81, codecThem.py:485,524-551)
    if (lane < fpw) {
        for (int i = 0; i < nTot; ++i) {
            const int band = i % nb, strm = i / nb;
            double v = 0.0;
            if (active) {
                const int sig = joint ? (msSwitch[f * nb + band] ? 2 + strm : strm) : 0;
                v = smr[(f * nsig + sig) * nb + band];
            }
            A.run[i * fpw + lane] = v;
            A.bits[i * fpw + lane] = 0;
        }
    }
    __syncthreads();
    const double r = (active && resIn) ? (double)resIn[f] : 0.0;
    double budget;
    if (joint) { budget = S.budgetJointPre + r; budget -= S.blkswA; budget -= S.blkswB; }   // codecThem.py:390-396
    else budget = S.budgetMono + r;                                                           // codecThem.py:308
    const double left = bitalloc_lane<NTOT>(A, nTot, S.maxMantBits, budget, lane, active, fpw);
    if (active) {
        resOut[f] = (int)left;                            // int(bitsLeft): truncation toward zero (bitalloc.py:155)
        for (int i = 0; i < nTot; ++i) bitAlloc[f * nTot + i] = A.bits[i * fpw + lane];
    }
}

// stage-level entry (mrc_bitalloc): independent problems with explicit budgets, shared nLines
__global__ __launch_bounds__(kWave) void bitalloc_cases_kernel(int64_t nCases, int nBands, int maxMantBits,
                                                               const int* __restrict__ nLines,
                                                               const double* __restrict__ budget,
                                                               const double* __restrict__ smr, int* __restrict__ bitsOut,
                                                               int* __restrict__ left, double* __restrict__ smrAfter) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const AllocSlabs A = alloc_slabs(lds, nBands, kWave);
    const int64_t c = (int64_t)blockIdx.x * kWave + lane;
    const bool active = c < nCases;
    for (int i = lane; i < nBands; i += kWave) A.nLines[i] = nLines[i];
    for (int i = 0; i < nBands; ++i) {
        A.run[i * kWave + lane] = active ? smr[c * nBands + i] : 0.0;
        A.bits[i * kWave + lane] = 0;
    }
    __syncthreads();
    const double l = bitalloc_lane(A, nBands, maxMantBits, active ? budget[c] : 0.0, lane, active);
    if (active) {
        left[c] = (int)l;
        for (int i = 0; i < nBands; ++i) bitsOut[c * nBands + i] = A.bits[i * kWave + lane];
        // bitalloc.py:132-151 updates the caller's SMR array in place: the running values after the loop
        if (smrAfter) for (int i = 0; i < nBands; ++i) smrAfter[c * nBands + i] = A.run[i * kWave + lane];
    }
}

// ------------------------------------------------------------------------------------------------
// scale factors + mantissas: one wavefront per (frame, stream)
// ------------------------------------------------------------------------------------------------
// LONGJ: -1 any shape, joint or not at run time; 0 / 1: the long block (1024 lines), independent / joint channels fixed at
// compile time (the four iterations of the line loop unroll, the stream arithmetic folds)
template <class OutT, int LONGJ>                       // int32 plane, or uint16 (codes are at most 16 bits wide: codecThem.py:292-293)
__global__ __launch_bounds__(kWave) void quantize_kernel(DevShape S, int jointArg, const double* __restrict__ lines,
                                                         const int* __restrict__ oscale,
                                                         const double* __restrict__ bandPeak,
                                                         const int* __restrict__ msSwitch,
                                                         const int* __restrict__ bitAlloc,
                                                         int* __restrict__ scaleFactor, OutT* __restrict__ mantissa,
                                                         int vecOk /* lines and mantissa planes 16-byte aligned */) {
    __shared__ unsigned int sInfo[kMaxBands];            // per band: bits | scale factor << 8 | overall scale << 16 | signal << 24
    __shared__ __attribute__((aligned(4))) unsigned char sBand[kBandLds];   // band of every line (copy of S.bandOfLine)
    const int lane = threadIdx.x;
    const int joint = LONGJ < 0 ? jointArg : LONGJ;
    const int nstream = joint ? 2 : 1, nsig = joint ? 4 : 1;
    const int64_t f = blockIdx.x / nstream;
    const int strm = blockIdx.x % nstream;
    const int M = LONGJ < 0 ? S.halfN : 1024, nb = S.nBands;
    const double* X = lines + f * nsig * M;
    const int* osc = oscale + f * nsig;
    // the line -> band map goes to LDS with the other per-band values: the loop below then has ONE global round
    // trip per batch of lines (the lines themselves) instead of two dependent ones per line
    const bool bandInLds = M <= kBandLds;
    if (bandInLds)
        for (int k = 4 * lane; k < M; k += 4 * kWave)
            *reinterpret_cast<unsigned int*>(sBand + k) = *reinterpret_cast<const unsigned int*>(S.bandOfLine + k);
    if (lane < nb) {
        const int sig = joint ? (msSwitch[f * nb + lane] ? 2 + strm : strm) : 0;
        const int ba = bitAlloc[(f * nstream + strm) * nb + lane];
        // codecThem.py:346-347: ScaleFactor(max |scaled line| of the band, nScaleBits, nMantBits = bitAlloc);
        // scaling by 2^overallScale is exact, so max and scale commute
        const double peak = ldexp(bandPeak[(f * nsig + sig) * nb + lane], osc[sig]);
        const int sf = scale_factor_dev(peak, S.nScaleBits, ba);
        sInfo[lane] = (unsigned)ba | ((unsigned)sf << 8) | ((unsigned)osc[sig] << 16) | ((unsigned)sig << 24);
        scaleFactor[(f * nstream + strm) * nb + lane] = sf;
    }
    __syncthreads();
    OutT* out = mantissa + (f * nstream + strm) * M;
    auto code_of = [&](double x, unsigned info) -> OutT {   // codecThem.py:348-349
        const int ba = (int)(info & 0xff);
        return (OutT)(ba ? mantissa_dev(ldexp(x, (int)((info >> 16) & 0xff)), (int)((info >> 8) & 0xff), S.nScaleBits, ba) : 0);
    };
    if ((LONGJ >= 0 || vecOk) && bandInLds && !(M & 3)) {       // (LONGJ: the launcher has checked the alignment)
        // a lane takes FOUR CONSECUTIVE lines: one LDS word for their bands, 32 contiguous bytes of lines (one signal: the
        // four lines of a lane lie in one band almost always), one 8- or 16-byte store of the four codes
        for (int k = 4 * lane; k < M; k += 4 * kWave) {
            const unsigned bands = *reinterpret_cast<const unsigned int*>(sBand + k);
            unsigned info[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) info[u] = sInfo[(bands >> (8 * u)) & 0xff];
            double x[4] = {0.0, 0.0, 0.0, 0.0};
            const bool anyBits = ((info[0] | info[1] | info[2] | info[3]) & 0xff) != 0;   // lines of bands without bits are not read
            const bool oneSignal = ((info[0] ^ info[3]) >> 24) == 0 && ((info[1] ^ info[2]) >> 24) == 0 && ((info[0] ^ info[1]) >> 24) == 0;
            if (anyBits) {
                if (oneSignal) {
                    const double* src = X + (int64_t)(info[0] >> 24) * M + k;
                    const double2 p = *reinterpret_cast<const double2*>(src), q = *reinterpret_cast<const double2*>(src + 2);
                    x[0] = p.x; x[1] = p.y; x[2] = q.x; x[3] = q.y;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = X[(int64_t)(info[u] >> 24) * M + k + u];
                }
            }
            OutT c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = code_of(x[u], info[u]);
            if (sizeof(OutT) == 2) {
                uint2 w;
                w.x = (unsigned)(unsigned short)c[0] | ((unsigned)(unsigned short)c[1] << 16);
                w.y = (unsigned)(unsigned short)c[2] | ((unsigned)(unsigned short)c[3] << 16);
                *reinterpret_cast<uint2*>(out + k) = w;
            } else {
                *reinterpret_cast<int4*>(out + k) = make_int4((int)c[0], (int)c[1], (int)c[2], (int)c[3]);
            }
        }
    } else {
        constexpr int kBatch = 4;                        // lines in flight per lane
        for (int k0 = lane; k0 < M; k0 += kWave * kBatch) {
            unsigned info[kBatch];
            double x[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int k = min(k0 + u * kWave, M - 1);
                info[u] = sInfo[bandInLds ? sBand[k] : S.bandOfLine[k]];
                x[u] = (info[u] & 0xff) ? X[(int64_t)(info[u] >> 24) * M + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int k = k0 + u * kWave;
                if (k < M) out[k] = code_of(x[u], info[u]);
            }
        }
    }
}

}  // namespace

size_t alloc_workspace_bytes(const DevShape& S, int64_t nFrames, int joint) {
    return (size_t)nFrames * (joint ? 4 : 1) * S.nBands * sizeof(double);
}

hipError_t launch_alloc_quant(const DevShape& S, int64_t nFrames, int joint, const double* lines, const int* oscale,
                              const double* smr, const int* resIn, int* msSwitch, int* bitAlloc, int* scaleFactor,
                              void* mantissa, int mantFmt, int* resOut, double* bandPeakWs, bool peaksReady, bool msReady,
                              hipEvent_t* ev /* null, or 2 events: after band_stats, after bitalloc */, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    const int nTot = (joint ? 2 : 1) * S.nBands;
    // the per-band peaks come from smr_kernel on the full path; band_stats_kernel only serves the stage entry points
    if (!peaksReady)
        hipLaunchKernelGGL(band_stats_kernel, dim3((unsigned)nFrames), dim3(kWave), 0, st, S, joint, 1, lines, msSwitch,
                           bandPeakWs);
    if (joint && !msReady) {                             // on the UNSCALED L / R lines (codecThem.py:436)
        hipError_t e = launch_ms_switch(nFrames, S.nBands, S.msLeaves, S.msInternal, S.msPlan, lines, lines + S.halfN,
                                        4 * (int64_t)S.halfN, S.halfN, msSwitch, st);
        if (e != hipSuccess) return e;
    }
    if (ev) (void)hipEventRecord(ev[0], st);
    // frames per wave.  Spreading a launch over more, emptier waves (fpw = 8 .. 32) was tried to hide the latency of the
    // loop's dependent LDS reads: 2-3x SLOWER (mono 0.16 -> 0.30 ms, joint 0.52 -> 1.51 ms per 131 072 / 65 536 frames) --
    // the loop is bound by instruction issue, not by latency, so full waves it is.
    const int fpw = kWave;
    const size_t lds = alloc_lds_bytes(nTot, fpw);
    const dim3 bgrid((unsigned)((nFrames + fpw - 1) / fpw));
    // the common band counts (25 bands of a long block at 44.1 / 48 kHz, one or two streams) as compile-time constants
    // (9 / 18: the short and transition blocks)
    if (nTot == 9) hipLaunchKernelGGL(bitalloc_kernel<9>, bgrid, dim3(kWave), lds, st, S, joint, nFrames, fpw, smr, msSwitch, resIn, bitAlloc, resOut);
    else if (nTot == 18) hipLaunchKernelGGL(bitalloc_kernel<18>, bgrid, dim3(kWave), lds, st, S, joint, nFrames, fpw, smr, msSwitch, resIn, bitAlloc, resOut);
    else if (nTot == 25) hipLaunchKernelGGL(bitalloc_kernel<25>, bgrid, dim3(kWave), lds, st, S, joint, nFrames, fpw, smr, msSwitch, resIn, bitAlloc, resOut);
    else if (nTot == 50) hipLaunchKernelGGL(bitalloc_kernel<50>, bgrid, dim3(kWave), lds, st, S, joint, nFrames, fpw, smr, msSwitch, resIn, bitAlloc, resOut);
    else hipLaunchKernelGGL(bitalloc_kernel<0>, bgrid, dim3(kWave), lds, st, S, joint, nFrames, fpw, smr, msSwitch, resIn, bitAlloc, resOut);
    if (ev) (void)hipEventRecord(ev[1], st);
    const int vecOk = !((reinterpret_cast<uintptr_t>(lines) | reinterpret_cast<uintptr_t>(mantissa)) & 15);
    const dim3 qgrid((unsigned)(nFrames * (joint ? 2 : 1)));
    const int longj = (S.halfN == 1024 && vecOk) ? (joint ? 1 : 0) : -1;
#define MRC_Q_LAUNCH(TY, LJ)                                                                                         \
    hipLaunchKernelGGL((quantize_kernel<TY, LJ>), qgrid, dim3(kWave), 0, st, S, joint, lines, oscale, bandPeakWs,    \
                       msSwitch, bitAlloc, scaleFactor, (TY*)mantissa, vecOk)
#define MRC_Q_PICK(TY) do { if (longj == 0) MRC_Q_LAUNCH(TY, 0); else if (longj == 1) MRC_Q_LAUNCH(TY, 1);            \
                            else MRC_Q_LAUNCH(TY, -1); } while (0)
    if (mantFmt == MRC_MANTISSA_I16) MRC_Q_PICK(unsigned short);
    else MRC_Q_PICK(int);
#undef MRC_Q_PICK
#undef MRC_Q_LAUNCH
    return hipGetLastError();
}

hipError_t launch_bitalloc_cases(int64_t nCases, int nBands, int maxMantBits, const int* nLines, const double* budget,
                                 const double* smr, int* bits, int* left, double* smrAfter, hipStream_t st) {
    if (nCases <= 0) return hipSuccess;
    const size_t lds = alloc_lds_bytes(nBands, kWave);
    hipLaunchKernelGGL(bitalloc_cases_kernel, dim3((unsigned)((nCases + kWave - 1) / kWave)), dim3(kWave), lds, st,
                       nCases, nBands, maxMantBits, nLines, budget, smr, bits, left, smrAfter);
    return hipGetLastError();
}

}  // namespace mrc
