// sensitivity_kernel -- a diagnostic pass over the intermediate results of an encode (round 4): how many of its INTEGER
// decisions were taken within a guard band of floating-point rounding.
//
// Every integer the path emits is a floor / compare of float64 values that went through FFTs, log10, atan and 2^x whose
// last bits differ between this implementation and the reference's NumPy: the lines by up to
// ~2.6e-13 of the block peak, thresholds / SMRs by up to ~1e-10 dB.  A decision can only come out differently where the
// deciding value sits that close to its edge -- a ~1e-9-per-frame event, which at 10^7 frames is no longer "never".  This
// kernel counts those places, so that "bit-exact" can be stated per call as "no decision was near an edge" (or: these
// many were, in these frames -- re-encode them with MRC_OPT_EXACT_SPREAD and compare, cli --certify).
//
// Categories (mrc_hip.h MRC_SENS_*):
//   QUANT   codecThem.py:346-349 / quantize.py:12-38,294-322: a mantissa whose code t = ((2^R - 1)|x| + 1)/2 lies within
//           (2^R - 1)/2 x kLineGuard of a multiple of 2^(15 - scale) (the code's truncation edge), and a band whose
//           scale factor is decided by a peak code within the same distance of a power of two (quantize.py:114-146);
//           x = the scaled line (peak of the block in [1/2, 1) unless the overall scale is capped), kLineGuard = 4e-13
//   BITALLOC bitalloc.py:132-151: two bands (of the 25 / 50 the allocation runs over) whose SMRs differ by a multiple of
//           6 dB to within kDbGuard = 1e-9 dB: their running values S - 12 - 6 j tie at some step, and np.argmax's order
//           between them hangs on the last bits
//   MS      ms_stereo.py:5-27: a band whose sum|L^2 - R^2| is within kMsGuard = 1e-12 (relative) of 0.8 sum|L^2 + R^2|
//   (PEAK: psychoac.py:162's strict comparisons, and NODES: chunks the slope-node evaluation sent back to the sorted
//    sweep, are counted inside smr_kernel, which holds the spectrum.)
// One 256-thread workgroup per frame; runs only when MRC_OPT_SENSITIVITY is set.
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
namespace {

constexpr double kLineGuard = 4e-13;
constexpr double kDbGuard = 1e-9;
constexpr double kMsGuard = 1e-12;

__global__ __launch_bounds__(256) void sensitivity_kernel(DevShape S, int joint, int64_t nFrames,
                                                          const double* __restrict__ lines, const int* __restrict__ oscale,
                                                          const double* __restrict__ smr, const double* __restrict__ bandPeak,
                                                          const int* __restrict__ msSwitch, const int* __restrict__ bitAlloc,
                                                          const int* __restrict__ scaleFactor,
                                                          unsigned long long* __restrict__ sens,
                                                          unsigned char* __restrict__ frameFlags) {
    __shared__ double sD[kMaxBands], sS[kMaxBands], sKey[2 * kMaxBands];
    __shared__ unsigned sInfo[2 * kMaxBands];
    __shared__ unsigned sCount[3];
    const int tid = threadIdx.x;
    const int64_t f = blockIdx.x;
    const double gs = __longlong_as_double((long long)sens[7]);       // guard scale: 1, or 1e8 in the tests' loose mode
    const int nb = S.nBands, M = S.halfN, nsig = joint ? 4 : 1, nstream = joint ? 2 : 1, nTot = nstream * nb;
    const double* X = lines + f * nsig * M;
    if (tid < kMaxBands) { sD[tid] = 0.0; sS[tid] = 0.0; }
    if (tid < 3) sCount[tid] = 0;
    if (tid < nTot) {
        const int band = tid % nb, strm = tid / nb;
        const int sig = joint ? (msSwitch[f * nb + band] ? 2 + strm : strm) : 0;
        const int ba = bitAlloc[(f * nstream + strm) * nb + band], sf = scaleFactor[(f * nstream + strm) * nb + band];
        sInfo[tid] = (unsigned)ba | ((unsigned)sf << 8) | ((unsigned)oscale[f * nsig + sig] << 16) | ((unsigned)sig << 24);
        sKey[tid] = smr[(f * nsig + sig) * nb + band];
    }
    __syncthreads();
    unsigned mine0 = 0, mine1 = 0, mine2 = 0;
    // ---- MS: band energies of the L / R lines (plain sums: a margin test, not the decision itself)
    if (joint) {
        for (int k = tid; k < M; k += 256) {
            const double l = X[k], r = X[M + k];
            const int b = S.bandOfLine[k];
            atomicAdd(&sD[b], fabs(l * l - r * r));
            atomicAdd(&sS[b], fabs(l * l + r * r));
        }
    }
    // ---- QUANT: every coded line against the truncation edges of its code
    for (int strm = 0; strm < nstream; ++strm)
        for (int k = tid; k < M; k += 256) {
            const unsigned info = sInfo[strm * nb + S.bandOfLine[k]];
            const int ba = (int)(info & 0xff);
            if (!ba) continue;
            const int sf = (int)((info >> 8) & 0xff), osc = (int)((info >> 16) & 0xff), sig = (int)(info >> 24);
            const double x = fabs(ldexp(X[(int64_t)sig * M + k], osc));
            if (x >= 1.0) continue;                       // saturated code: no edge nearby
            const int nBits = ((1 << S.nScaleBits) - 1) + ba;
            const double m = (double)((1LL << nBits) - 1);
            const double t = (m * x + 1.0) / 2.0;
            int shift = ((1 << S.nScaleBits) - 1) - sf;
            if (shift < 0) shift = 0;
            const double step = ldexp(1.0, shift);
            const double q = t / step;
            const double dist = fabs(q - rint(q)) * step;             // distance of t to the nearest multiple of 2^shift
            if (dist <= 0.5 * m * (kLineGuard * gs)) ++mine0;
        }
    __syncthreads();
    // scale factors: the band peak's code against the powers of two that separate leading-zero counts
    if (tid < nTot) {
        const unsigned info = sInfo[tid];
        const int ba = (int)(info & 0xff), osc = (int)((info >> 16) & 0xff), sig = (int)(info >> 24);
        const int band = tid % nb;
        const double peak = fabs(ldexp(bandPeak[(f * nsig + sig) * nb + band], osc));
        const int nBits = ((1 << S.nScaleBits) - 1) + ba;
        if (ba && peak < 1.0 && peak > 0.0) {
            const double m = (double)((1LL << nBits) - 1);
            const double t = (m * peak + 1.0) / 2.0;
            if (t >= 1.0) {
                const int e = ilogb(t);
                const double lo = ldexp(1.0, e), hi = ldexp(1.0, e + 1);
                if (fmin(t - lo, hi - t) <= 0.5 * m * (kLineGuard * gs)) ++mine0;
            }
        }
        // ---- BITALLOC: SMRs a multiple of 6 dB apart
        const double a = sKey[tid];
        for (int j = tid + 1; j < nTot; ++j) {
            const double d = a - sKey[j];
            if (!(fabs(d) < 1e6)) continue;                // (a band whose SMR is the "no line" marker)
            const double r = d - 6.0 * rint(d / 6.0);
            if (fabs(r) <= kDbGuard * gs) ++mine1;
        }
    }
    if (joint && tid < nb) {
        const double lhs = sD[tid], rhs = 0.8 * sS[tid];
        if (fabs(lhs - rhs) <= (kMsGuard * gs) * sS[tid] && sS[tid] > 0.0) ++mine2;
    }
    if (mine0) atomicAdd(&sCount[0], mine0);
    if (mine1) atomicAdd(&sCount[1], mine1);
    if (mine2) atomicAdd(&sCount[2], mine2);
    __syncthreads();
    if (tid == 0) {
        if (sCount[0]) atomicAdd(&sens[0], (unsigned long long)sCount[0]);
        if (sCount[1]) atomicAdd(&sens[1], (unsigned long long)sCount[1]);
        if (sCount[2]) atomicAdd(&sens[2], (unsigned long long)sCount[2]);
        atomicAdd(&sens[5], 1ull);                       // frames examined
        const unsigned char fl = (unsigned char)((sCount[0] ? 1 : 0) | (sCount[1] ? 2 : 0) | (sCount[2] ? 4 : 0));
        if (frameFlags && fl) frameFlags[f] |= fl;
    }
}

}  // namespace

hipError_t launch_sensitivity(const DevShape& S, int64_t nFrames, int joint, const double* lines, const int* oscale,
                              const double* smr, const double* bandPeak, const int* msSwitch, const int* bitAlloc,
                              const int* scaleFactor, unsigned long long* sens, unsigned char* frameFlags, hipStream_t st) {
    if (nFrames <= 0 || !sens) return hipSuccess;
    hipLaunchKernelGGL(sensitivity_kernel, dim3((unsigned)nFrames), dim3(256), 0, st, S, joint, nFrames, lines, oscale, smr,
                       bandPeak, msSwitch, bitAlloc, scaleFactor, sens, frameFlags);
    return hipGetLastError();
}

}  // namespace mrc
