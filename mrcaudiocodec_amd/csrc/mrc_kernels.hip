// gfx950 kernels of the encode hot path, generic in the block shape (a,b):
//   mdct_kernel        window.py:104-121 + mdct.py:63-76 + codecThem.py:321-322
//   (smr_kernel lives in mrc_kernels_smr.hip)
//   (bit allocation / quantisation kernels live in mrc_kernels_alloc.hip)
// All arithmetic is binary64.  The file is compiled with -ffp-contract=off: wherever the reference's
// operation order decides an integer result (quantiser, bit allocation, SMR) the same separate
// multiplies/adds are issued; fma() is used only where written explicitly.
//
// This is the shape-generic path (any a,b whose N/4 and N/2 factor into 2s and 3s).  Faster
// specialisations for the long block live in mrc_kernels_long.hip.
#include "mrc_device.hpp"

#include <algorithm>
#include <map>
#include <mutex>

namespace mrc {
using namespace dev;
namespace {

// ------------------------------------------------------------------------------------------------
// MDCT: one workgroup per (frame, signal)
// ------------------------------------------------------------------------------------------------
template <class SampleT, int NT>
__global__ __launch_bounds__(NT) void mdct_kernel(DevShape S, int nsig, const SampleT* __restrict__ chL,
                                                        const SampleT* __restrict__ chR, int64_t stride,
                                                        const int64_t* __restrict__ offsets,
                                                        const double* __restrict__ win,
                                                        double* __restrict__ lines, int* __restrict__ oscale) {
    extern __shared__ double smem[];
    __shared__ double red[NT / kWave];
    const int tid = threadIdx.x;
    const int N = S.N, M = S.halfN, Q = S.Q;
    const int64_t f = blockIdx.x / nsig;
    const int sig = blockIdx.x % nsig;
    const int64_t off = offsets ? offsets[f] : f * stride;
    double* y = smem;                                   // [N]   windowed, phase-shifted block; later the output lines
    double2* A = (double2*)(smem + N);                  // [Q]
    double2* B = A + Q;                                 // [Q]

    // window (window.py:104-121) + signed circular shift by (b-a)/4: the transform kernel is
    // anti-periodic in N, so X(n0=(b+1)/2) of x equals the standard-phase MDCT of the shifted block.
    for (int n = tid; n < N; n += NT) {
        double v = load_signal(chL, chR, off + n, sig);
        if (win) v = v * win[n];
        int m = n + S.shift;
        if (m < 0) { m += N; v = -v; }
        else if (m >= N) { m -= N; v = -v; }
        y[m] = v;
    }
    __syncthreads();
    // fold N -> N/2 (DCT-IV input u) and pack pairs into Q complex points with the pre-twiddle
    for (int n = tid; n < Q; n += NT) {
        const int j0 = 2 * n, j1 = M - 1 - 2 * n, h = Q;
        double u0 = (j0 < h) ? (-y[3 * h - 1 - j0] - y[3 * h + j0]) : (y[j0 - h] - y[3 * h - 1 - j0]);
        double u1 = (j1 < h) ? (-y[3 * h - 1 - j1] - y[3 * h + j1]) : (y[j1 - h] - y[3 * h - 1 - j1]);
        A[n] = cmul(make_double2(u0, u1), S.pre[n]);
    }
    __syncthreads();
    double2* T = fft_lds_global<NT>(A, B, Q, S.radQ, S.nRadQ, S.wQ, tid);
    // post-twiddle; y is free again (all reads of it happened before the FFT's first barrier)
    for (int k = tid; k < Q; k += NT) {
        double2 c = cmul(T[k], S.post[k]);
        y[2 * k] = S.twoOverN * c.x;
        y[M - 1 - 2 * k] = S.twoOverN * (-c.y);
    }
    __syncthreads();
    double peak = 0.0;
    double* dst = lines + ((int64_t)blockIdx.x) * M;
    for (int k = tid; k < M; k += NT) {
        double v = y[k];
        dst[k] = v;
        peak = fmax(peak, fabs(v));
    }
    peak = wave_max(peak);
    if ((tid & (kWave - 1)) == 0) red[tid / kWave] = peak;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / kWave; ++w) peak = fmax(peak, red[w]);
        oscale[blockIdx.x] = scale_factor_dev(peak, S.nScaleBits, 5);       // codecThem.py:322 (nMantBits default)
    }
}

// ------------------------------------------------------------------------------------------------
// MDCT of the short (128 + 128) and transition (1024 + 128, 128 + 1024) blocks of the reference's block switching: ONE
// WAVEFRONT per group of U units, no workgroup barrier inside the loop.  Same transform and formulas as mdct_kernel above.
// What mdct_kernel costs on these shapes is instructions, not bytes: its passes run Q / R butterflies on NT threads (16 of 64
// lanes busy for a short block's radix-4 passes, 72 of 256 threads for a transition block), with run-time radices, strides
// and divisions -- ~1 500 instructions per wave and unit -- and every unit re-reads ~5 KB of window / twiddle constants
// through a chain of dependent loads.  Here
//   * U units share a wave so that its lanes are busy: four short blocks give 4 x 16 = 64 radix-4 butterflies per pass; a
//     transition block's 288-point FFT runs as 8 x 4 x 3 x 3 in 1 + 2 + 2 + 2 rounds of 64 lanes;
//   * every size is a template parameter (index splits and twiddle strides fold into immediates);
//   * the window values and pre- / post-twiddles of a lane's sample and point slots live in registers and the FFT's twiddle
//     table in LDS, loaded once per workgroup, which then walks `run` groups per wave;
//   * a unit's samples come as coalesced loads of 64 consecutive samples (N is a multiple of 64, so a load never straddles two
//     units and the unit of a slot is wave-uniform: offsets and signal ids stay scalar).
// ------------------------------------------------------------------------------------------------
struct TwLds {
    const double2* w;                                   // LDS, [n]
    __device__ __forceinline__ double2 operator()(int t, int /*nq*/) const { return w[t]; }
};
// LDS traffic between lanes of ONE wave: order the wave's own DS operations and keep the compiler from moving LDS accesses
// across this point
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// one Stockham pass of radix R over U transforms of Q points each (P = product of the radices already done), a butterfly
// per lane and round
template <int R, int Q, int P, int U>
__device__ __forceinline__ void wave_pass(const double2* in, double2* out, const double2* wl, int lane) {
    constexpr int T = Q / R, tws = Q / (P * R), total = U * T;
#pragma unroll
    for (int b0 = 0; b0 < total; b0 += kWave) {
        const int b = b0 + lane;
        if (b0 + kWave <= total || b < total) {
            const int unit = b / T, i = b % T;
            const int k = i % P, j = (i / P) * (P * R) + k;
            const double2* src = in + unit * Q;
            double2* dst = out + unit * Q;
            double2 u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = src[i + r * T];
            if (P != 1) {
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], wl[k * r * tws]);
            }
            butterfly<R>(u);
#pragma unroll
            for (int q = 0; q < R; ++q) dst[j + q * P] = u[q];
        }
    }
}
// the passes of the two sizes; returns the buffer that holds the natural-order result
template <int Q, int U>
__device__ __forceinline__ double2* wave_fft(double2* A, double2* B, const double2* wl, int lane) {
    static_assert(Q == 64 || Q == 288, "sizes of the reference's short and transition blocks");
    if (Q == 64) {
        wave_pass<4, Q, 1, U>(A, B, wl, lane);
        wave_sync();
        wave_pass<4, Q, 4, U>(B, A, wl, lane);
        wave_sync();
        wave_pass<4, Q, 16, U>(A, B, wl, lane);
        wave_sync();
        return B;
    }
    wave_pass<8, Q, 1, U>(A, B, wl, lane);
    wave_sync();
    wave_pass<4, Q, 8, U>(B, A, wl, lane);
    wave_sync();
    wave_pass<3, Q, 32, U>(A, B, wl, lane);
    wave_sync();
    wave_pass<3, Q, 96, U>(B, A, wl, lane);
    wave_sync();
    return A;
}

// Four per-lane partial maxima (one per unit of a group) -> every lane of row r (lanes 16 r .. 16 r + 15) ends with the
// wave-wide maximum of value r: two rounds of permlane swaps fold the four registers into one whose rows belong to the four
// values, then one reduction inside the rows -- a quarter of four separate wave reductions.
__device__ __forceinline__ double wave_max4(double p0, double p1, double p2, double p3) {
    auto swap = [](double a, double b, bool half, double* x, double* y) {
        const unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
        const unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
        const uint2v l = half ? __builtin_amdgcn_permlane32_swap(al, bl, false, false)
                              : __builtin_amdgcn_permlane16_swap(al, bl, false, false);
        const uint2v h = half ? __builtin_amdgcn_permlane32_swap(ah, bh, false, false)
                              : __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
        *x = __hiloint2double((int)h.x, (int)l.x);
        *y = __hiloint2double((int)h.y, (int)l.y);
    };
    double x, y;
    swap(p0, p2, true, &x, &y);                         // x = {p0 lower half, p2 lower half}, y = {p0 upper, p2 upper}
    const double m02 = fmax(x, y);                      // lanes 0-31: value 0 over both halves; lanes 32-63: value 2
    swap(p1, p3, true, &x, &y);
    const double m13 = fmax(x, y);                      // lanes 0-31: value 1; lanes 32-63: value 3
    swap(m02, m13, false, &x, &y);                      // x = {m02 row 0, m13 row 0, m02 row 2, m13 row 2}, y = the odd rows
    double q = fmax(x, y);                              // row r: value r
    q = fmax(q, dpp_move<0xB1>(q));
    q = fmax(q, dpp_move<0x4E>(q));
    q = fmax(q, dpp_move<0x141>(q));
    return fmax(q, dpp_move<0x128>(q));
}

constexpr int kMdctWaves = 4;                           // waves per workgroup, each with its own groups of units
constexpr int kMdctMaxRun = 16;                         // groups per wave, at most
// K64 != 0 (transition blocks, N = 1152, shift = -+224): the fold N -> N/2 happens in REGISTERS.  The two samples that fold
// into one DCT-IV input are n and (64 K64 + 63 - n) mod N (K64 = 15 for 1024 + 128, 1 for 128 + 1024), i.e. chunk c of 64
// samples pairs with chunk (K64 - c) mod 18 read backwards: the second chunk of every pair is loaded with the lane order
// reversed, the partners sit in the same lane, their signs (window wrap and fold) ride on the lane's window values, and only
// the N/2 sums travel through LDS.  Same products, same two-term sums as the staged form (x - y = x + (-y) exactly): identical
// lines.  9 instead of 13.5 KiB of LDS per wave and no shift bookkeeping: three workgroups per CU instead of two.
template <int K64> __host__ __device__ constexpr int fold_partner(int c) { return (K64 - c + 18) % 18; }
template <class SampleT, int N, int U, int NSIG, bool SHIFT0, int K64>
__global__ __launch_bounds__(kWave * kMdctWaves, K64 ? 3 : 2) void mdct_wave_kernel(
    DevShape S, int64_t nUnits, int64_t nWaves, const SampleT* __restrict__ chL, const SampleT* __restrict__ chR, int64_t stride,
    const int64_t* __restrict__ offsets, double* __restrict__ lines, int* __restrict__ oscale) {
    constexpr int Q = N / 4, M = N / 2;
    constexpr int kSlots = U * N / kWave;                // sample slots per lane: s = lane + 64 c, unit c / (N / 64)
    constexpr int kSlotsPerUnit = N / kWave;
    constexpr int kPoints = (U * Q + kWave - 1) / kWave; // FFT points per lane: g = lane + 64 c
    // doubles per wave: y [U][N] (windowed samples; dead after the fold, then the FFT's second buffer B [U][Q] complex and,
    // for the sizes whose result ends in A, the staging of the output lines) | A [U][Q] complex (and the staging otherwise)
    constexpr bool RF = K64 != 0;
    constexpr int kWaveLds = RF ? 2 * U * M : U * N + 2 * U * Q;
    static_assert(N % kWave == 0 && ((U * Q) % kWave == 0 || U == 1), "a slot never straddles two units");
    static_assert(!RF || (U == 1 && N == 1152 && !SHIFT0), "register fold: one transition block per wave");
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* y = smem + wave * kWaveLds;
    double2* A = reinterpret_cast<double2*>(y + (RF ? U * M : U * N));
    double2* B = reinterpret_cast<double2*>(y);
    double2* wl = reinterpret_cast<double2*>(smem + kMdctWaves * kWaveLds);          // [Q] e^{-2 pi i t/Q}
    __shared__ long long sOff[kMdctWaves][kMdctMaxRun * U];                          // sample offset of every unit of a wave's run
    // the groups are dealt to the launch's nWaves wavefronts in contiguous runs that differ by at most one group (the host
    // sizes nWaves so that the workgroups fill the chip a whole number of times and no run exceeds kMdctMaxRun)
    const int64_t nGroups = (nUnits + U - 1) / U;
    const int64_t waveId = (int64_t)blockIdx.x * kMdctWaves + wave;
    const int64_t firstGroup = waveId < nWaves ? waveId * nGroups / nWaves : 0;
    const int run = waveId < nWaves ? (int)((waveId + 1) * nGroups / nWaves - firstGroup) : 0;   // (0: a wave beyond the last)
    // the sample offsets of all units of this wave's run, fetched at once (a scalar load per group inside the loop would put
    // a memory round trip in front of every group's sample loads)
    if (lane < run * U) {
        const int64_t unit = min(firstGroup * U + lane, nUnits - 1);                 // (the tail repeats the last unit)
        const int64_t f = NSIG == 1 ? unit : unit / NSIG;
        sOff[wave][lane] = offsets ? offsets[f] : f * stride;
    }
    wave_sync();
    // raw samples of the group in flight: requested one group ahead, converted when their group starts.  Coalesced: 64
    // consecutive samples of one unit per load; a joint group needs both channels (its units are consecutive signals of the
    // same frames).
    SampleT rawL[kSlots], rawR[NSIG == 1 ? 1 : kSlots];
    auto request = [&](int it) {
#pragma unroll
        for (int c = 0; c < kSlots; ++c) {
            const int ln = (RF && c > fold_partner<K64>(c)) ? kWave - 1 - lane : lane;
            const int64_t i = sOff[wave][it * U + c / kSlotsPerUnit] + ln + kWave * (c % kSlotsPerUnit);
            rawL[c] = chL[i];
            if (NSIG != 1) rawR[c] = chR[i];
        }
    };
    if (run > 0) request(0);                            // in flight while the constants below are fetched
    double2* preL = wl + Q;                             // register fold: pre- / post-twiddles in LDS too ([Q] each): the 40
    double2* postL = preL + Q;                          // registers they would take are what keeps three waves on a SIMD
    for (int t = threadIdx.x; t < Q; t += kWave * kMdctWaves) {
        wl[t] = S.wQ[t];
        if (RF) { preL[t] = S.pre[t]; postL[t] = S.post[t]; }
    }
    // lane constants: window value of every sample slot and where the sample goes after the signed circular shift by
    // (b-a)/4 (see mdct_kernel; bit 31: negated; SHIFT0: a = b, no shift); pre- and post-twiddle of every point slot
    double wv[kSlots];
    unsigned dst[SHIFT0 ? 1 : kSlots];
#pragma unroll
    for (int c = 0; c < kSlots; ++c) {
        if (RF) {
            // sample n of this lane's slot (second chunk of a pair: lanes reversed), where it lands after the signed circular
            // shift (m), which DCT-IV input it folds into (j) and with which sign (mdct_kernel's fold, by quarter of m)
            const bool first = c < fold_partner<K64>(c);
            const int n = (first ? lane : kWave - 1 - lane) + kWave * c;
            int m = n + S.shift;
            bool neg = false;
            if (m < 0) { m += N; neg = true; }
            else if (m >= N) { m -= N; neg = true; }
            int j;
            if (m < Q) j = m + Q;
            else if (m < 3 * Q) { j = 3 * Q - 1 - m; neg = !neg; }
            else { j = m - 3 * Q; neg = !neg; }
            wv[c] = neg ? -S.win[n] : S.win[n];
            dst[SHIFT0 ? 0 : c] = (unsigned)j;
            continue;
        }
        const int n = lane + kWave * (c % kSlotsPerUnit);
        wv[c] = S.win[n];
        if (!SHIFT0) {
            int m = n + S.shift;
            unsigned neg = 0;
            if (m < 0) { m += N; neg = 1u << 31; }
            else if (m >= N) { m -= N; neg = 1u << 31; }
            dst[SHIFT0 ? 0 : c] = (unsigned)(m + (c / kSlotsPerUnit) * N) | neg;
        }
    }
    double2 preR[RF ? 1 : kPoints], postR[RF ? 1 : kPoints];
#pragma unroll
    for (int c = 0; c < (RF ? 0 : kPoints); ++c) {
        const int n = min((lane + kWave * c) % Q, Q - 1);
        preR[c] = S.pre[n];
        postR[c] = S.post[n];
    }
    __syncthreads();                                    // twiddle table visible to all waves
    if (run == 0) return;                               // wave-uniform, after the workgroup's only barrier
    for (int it = 0; it < run; ++it) {
        const int64_t u0 = (firstGroup + it) * U;
        if (u0 >= nUnits) break;                        // wave-uniform
        // ---- A. convert, window, shift (register fold: ... and fold)
        double xs[RF ? kSlots : 1];
#pragma unroll
        for (int c = 0; c < kSlots; ++c) {
            const int sig = NSIG == 1 ? 0 : (int)(min(u0 + c / kSlotsPerUnit, nUnits - 1) % NSIG);
            double v;
            if (NSIG == 1 || sig == 0) v = sample_of(&rawL[c], 0);
            else if (sig == 1) v = sample_of(&rawR[NSIG == 1 ? 0 : c], 0);
            else {
                const double l = sample_of(&rawL[c], 0), r = sample_of(&rawR[NSIG == 1 ? 0 : c], 0);   // codecThem.py:363-364
                v = sig == 2 ? (l + r) / 2.0 : (l - r) / 2.0;
            }
            const double x = v * wv[c];
            if (RF) { xs[RF ? c : 0] = x; continue; }
            if (SHIFT0) y[(c / kSlotsPerUnit) * N + lane + kWave * (c % kSlotsPerUnit)] = x;
            else y[dst[SHIFT0 ? 0 : c] & 0x7fffffffu] = (dst[SHIFT0 ? 0 : c] >> 31) ? -x : x;
        }
        if (RF) {
#pragma unroll
            for (int c = 0; c < kSlots; ++c)
                if (c < fold_partner<K64>(c)) y[dst[SHIFT0 ? 0 : c]] = xs[RF ? c : 0] + xs[RF ? fold_partner<K64>(c) : 0];
        }
        if (it + 1 < run && u0 + U < nUnits) request(it + 1);          // in flight while this group is transformed
        wave_sync();
        // ---- B. fold N -> N/2 (DCT-IV input u) and pack pairs into Q complex points with the pre-twiddle
#pragma unroll
        for (int c = 0; c < kPoints; ++c) {
            const int g = lane + kWave * c;
            if ((c + 1) * kWave <= U * Q || g < U * Q) {
                const int ug = g / Q, n = g % Q;
                if (RF) {
                    A[g] = cmul(make_double2(y[2 * n], y[M - 1 - 2 * n]), preL[n]);
                    continue;
                }
                const double* yu = y + ug * N;
                const int j0 = 2 * n, j1 = M - 1 - 2 * n, h = Q;
                const double u0v = (j0 < h) ? (-yu[3 * h - 1 - j0] - yu[3 * h + j0]) : (yu[j0 - h] - yu[3 * h - 1 - j0]);
                const double u1v = (j1 < h) ? (-yu[3 * h - 1 - j1] - yu[3 * h + j1]) : (yu[j1 - h] - yu[3 * h - 1 - j1]);
                A[g] = cmul(make_double2(u0v, u1v), preR[RF ? 0 : c]);
            }
        }
        wave_sync();
        // ---- C. U transforms of Q points
        const double2* T = wave_fft<Q, U>(A, B, wl, lane);
        // ---- D. post-twiddle into the buffer the result is not in (U x M doubles): X[2k] = (2/N) Re, X[N/2-1-2k] = -(2/N) Im
        double* stage = (T == A) ? y : reinterpret_cast<double*>(A);
#pragma unroll
        for (int c = 0; c < kPoints; ++c) {
            const int g = lane + kWave * c;
            if ((c + 1) * kWave <= U * Q || g < U * Q) {
                const int ug = g / Q, k = g % Q;
                const double2 cc = cmul(T[g], RF ? postL[k] : postR[RF ? 0 : c]);
                double* xu = stage + ug * M;
                xu[2 * k] = S.twoOverN * cc.x;
                xu[M - 1 - 2 * k] = S.twoOverN * (-cc.y);
            }
        }
        wave_sync();
        // ---- E. lines out (512 / 1024 contiguous bytes per store), overall scale per unit (codecThem.py:321-322)
        if (U == 4 && M == 2 * kWave) {
            // four units of 128 lines: a 16-byte store per lane and unit, the four peaks reduced together
            double pk[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const double2 x = *reinterpret_cast<const double2*>(stage + uu * M + 2 * lane);
                if (u0 + uu < nUnits) *reinterpret_cast<double2*>(lines + (u0 + uu) * M + 2 * lane) = x;
                pk[uu] = fmax(fabs(x.x), fabs(x.y));
            }
            const double peak = wave_max4(pk[0], pk[1], pk[2], pk[3]);
            const int r = lane >> 4;
            if ((lane & 15) == 0 && u0 + r < nUnits) oscale[u0 + r] = scale_factor_dev(peak, S.nScaleBits, 5);
        } else
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            const int64_t unit = u0 + uu;
            if (unit >= nUnits) break;                  // wave-uniform
            double peak = 0.0;
            double* out = lines + unit * M;
#pragma unroll
            for (int c = 0; c < M / kWave; ++c) {
                const double x = stage[uu * M + lane + kWave * c];
                out[lane + kWave * c] = x;
                peak = fmax(peak, fabs(x));
            }
            peak = wave_max(peak);
            if (lane == 0) oscale[unit] = scale_factor_dev(peak, S.nScaleBits, 5);
        }
        wave_sync();                                    // staging read before the next group overwrites it
    }
}

// pcmfile.py:91-100: int16 PCM codes -> signed fractions (the map the int16 ingest paths apply on load)
__global__ void pcm_to_float_kernel(int64_t n, const short* __restrict__ pcm, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pcm16_to_frac(pcm[i]);
}

// quantize.py:12-38 / 61-87 elementwise: sign bit << (nBits-1) + magnitude code (x < 0.0 decides the sign: -0.0 is
// positive, as np.less has it)
__global__ void quantize_uniform_kernel(int64_t n, int nBits, const double* __restrict__ x, long long* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (x[i] < 0.0 ? (1LL << (nBits - 1)) : 0LL) + mag_code(fabs(x[i]), nBits);
}

// psychoac.py:27-29 elementwise
__global__ void bark_kernel(int64_t n, const double* __restrict__ f, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double q = f[i] / 7500.;
        out[i] = 13 * atan(0.76 * f[i] / 1000.) + 3.5 * atan(q * q);
    }
}

// ------------------------------------------------------------------------------------------------
// stage-level kernels for the parity tests against the reference's own modules
// ------------------------------------------------------------------------------------------------
__global__ void scale_factor_kernel(int64_t n, int nScaleBits, const double* __restrict__ v,
                                    const int* __restrict__ nMantBits, int* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = scale_factor_dev(v[i], nScaleBits, nMantBits[i]);
}

__global__ void mantissa_kernel(int64_t n, int nScaleBits, const double* __restrict__ x, const int* __restrict__ scale,
                                const int* __restrict__ nMantBits, int* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mantissa_dev(x[i], scale[i], nScaleBits, nMantBits[i]);
}

__global__ void window_kernel(int N, int64_t total, const double* __restrict__ win, const double* __restrict__ in,
                              double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = in[i] * win[i % N];
}

// scaled lines (X * 2^scale, codecThem.py:323) back to the unscaled lines the kernels keep; exact
__global__ void unscale_kernel(int halfN, int64_t total, const double* __restrict__ scaled,
                               const int* __restrict__ oscale, double* __restrict__ lines) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) lines[i] = ldexp(scaled[i], -oscale[i / halfN]);
}

// pacfileThem.py:1025-1056 (TransientDetector), the numeric part: every hop is high-pass filtered from a ZERO
// state by a cascade of second-order sections (scipy.signal.sosfilt's direct-form-II-transposed recurrence,
// same operation order) and the peak |y| of each short sub-block plus the peak of the whole hop are kept.
// One thread per (hop, channel): the recurrence is serial in time, hops are independent.
constexpr int kMaxSections = 16;
// NSEC > 0: the section count is a compile-time constant and the 5 x NSEC coefficients live in VECTOR registers (loaded
// through an address the compiler cannot prove uniform).  Left to itself the compiler keeps them behind scalar loads INSIDE
// the sample loop -- ten dependent scalar-memory round trips per sample for the reference's 20th-order filter, which is what
// the kernel's time used to be (0.87 ms per 131 072 hops); 100 scalar registers would not fit either.
// VEC: every hop starts on a 16-byte boundary: the samples come eight (int16) / two (double) per load, and the next group is
// requested before the current one is filtered.
// NSEC == 0: any section count up to kMaxSections, coefficients as the compiler pleases.
// UNIT: every section but the first has b0 == 1.0 exactly (what tf2sos / zpk2sos produce: the gain sits in the first
// section): their `b0 * x` is x -- the same value, one multiplication per section and sample less.
template <class SampleT, int NSEC, bool VEC, bool UNIT>
__global__ __launch_bounds__(kWave) void transient_peaks_kernel(int64_t nHops, int nCh, int hop, int nShort, int nSecArg,
                                                                 const double* __restrict__ sos,
                                                                 const SampleT* __restrict__ streams, int64_t chStride,
                                                                 double* __restrict__ peaks) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nHops * nCh) return;
    const int64_t h = t / nCh;
    const int ch = (int)(t % nCh);
    const SampleT* x = streams + ch * chStride + (h + 1) * hop;         // the stream starts with the prior hop
    const int nSub = hop / nShort;
    double* out = peaks + t * (nSub + 1);
    constexpr int kSec = NSEC ? NSEC : kMaxSections;
    const int nSec = NSEC ? NSEC : nSecArg;
    double z0[kSec], z1[kSec];
    double b0[kSec], b1[kSec], b2[kSec], a1[kSec], a2[kSec];
    int lane0 = 0;
    if (NSEC) asm volatile("" : "+v"(lane0));                        // an opaque zero: per-lane (vector) loads below
#pragma unroll
    for (int s = 0; s < kSec; ++s) {
        z0[s] = 0.0; z1[s] = 0.0;
        if (NSEC) {
            b0[s] = sos[6 * s + lane0]; b1[s] = sos[6 * s + 1 + lane0]; b2[s] = sos[6 * s + 2 + lane0];
            a1[s] = sos[6 * s + 4 + lane0]; a2[s] = sos[6 * s + 5 + lane0];
        }
    }
    // one sample through the cascade (scipy.signal.sosfilt's recurrence, same operation order)
    auto cascade = [&](double cur) -> double {
#pragma unroll
        for (int s = 0; s < kSec; ++s) {
            if (s < nSec) {
                const double c0 = NSEC ? b0[s] : sos[6 * s], c1 = NSEC ? b1[s] : sos[6 * s + 1];
                const double c2 = NSEC ? b2[s] : sos[6 * s + 2];
                const double d1 = NSEC ? a1[s] : sos[6 * s + 4], d2 = NSEC ? a2[s] : sos[6 * s + 5];
                const double y = (UNIT && s > 0) ? cur + z0[s] : c0 * cur + z0[s];
                z0[s] = (c1 * cur - d1 * y) + z1[s];
                z1[s] = c2 * cur - d2 * y;
                cur = y;
            }
        }
        return cur;
    };
    double all = 0.0;
    if (VEC) {
        constexpr int kGroup = 16 / sizeof(SampleT);                   // samples per 16-byte load
        typedef int int4n __attribute__((ext_vector_type(4)));
        const int4n* xv = reinterpret_cast<const int4n*>(x);
        const int groupsPerSub = nShort / kGroup, nGroups = hop / kGroup;
        int4n nxt = xv[0];
        int g = 0;
        for (int sb = 0; sb < nSub; ++sb) {
            double pk = 0.0;
            for (int gi = 0; gi < groupsPerSub; ++gi, ++g) {
                const int4n cur = nxt;
                nxt = xv[min(g + 1, nGroups - 1)];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    double v;
                    if (sizeof(SampleT) == 2) {
                        const int w = cur[j >> 1];
                        v = pcm16_to_frac((j & 1) ? (w >> 16) : (int)(short)(w & 0xffff));
                    } else {
                        v = __hiloint2double(cur[2 * j + 1], cur[2 * j]);
                    }
                    pk = fmax(pk, fabs(cascade(v)));
                }
            }
            out[sb] = pk;
            all = fmax(all, pk);
        }
    } else {
        for (int sb = 0; sb < nSub; ++sb) {
            double pk = 0.0;
            // (unrolling this loop by four, so that the cascades of neighbouring samples overlap, was measured: slower --
            // the section loop already fills the pipe)
            for (int n = 0; n < nShort; ++n) pk = fmax(pk, fabs(cascade(sample_of(x, sb * nShort + n))));
            out[sb] = pk;
            all = fmax(all, pk);
        }
    }
    out[nSub] = all;
}

// ms_stereo.py:53-67: masking-level-difference factor MLD = 10^(1.25 (1 - cos(pi min(z,15.5)/15.5) - 2.5)) and
// the cross-limited mid / side thresholds.  (The encoder's use of it is dead -- psychoac.py:205-210 -- the
// kernel exists so that the drop-in module keeps the reference's symbol.)
__global__ void stereo_masking_kernel(int64_t n, const double* __restrict__ mid, const double* __restrict__ side,
                                      const double* __restrict__ z, double* __restrict__ outMid,
                                      double* __restrict__ outSide) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double zc = fmin(z[i], 15.5);
    const double mld = pow(10.0, 1.25 * (1 - cos((M_PI / 15.5) * zc) - 2.5));
    const double m = mid[i], s = side[i];
    outMid[i] = fmax(m, fmin(s, mld * s));
    outSide[i] = fmax(s, fmin(m, mld * m));
}

// ms_stereo.py:5-27 for one block per wavefront: per band, sum|L^2 - R^2| < 0.8 sum|L^2 + R^2| with both sums rounded
// exactly as np.sum rounds them (pairwise summation, restated as the static tree of mrc::ms_plan).  A LEAF is a run
// of <= 128 lines: eight lanes hold NumPy's eight strided accumulators r[0..7], combine them in its fixed order
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) with three in-register exchanges and add the < 8 trailing elements one by one;
// runs of fewer than 8 lines are plain left-to-right sums.  Eight leaves are in flight per wave (a "round"; the plan lists
// the leaves longest first so that the rounds are even); the few internal nodes (bands of more than 128 lines) follow in
// tree order.
// No staging, and every load of a round in flight at once.  Earlier forms (a leaf walked step by step from global memory;
// the block staged through 16.5 KB of LDS per wave) all took 0.31-0.35 ms per 65 536 long blocks whatever their memory
// pattern: the kernel was bound by its chain of dependent round trips at 9-16 waves per CU.  Here a lane's <= 16 strided
// elements of a leaf are requested together (predicated loads: idle lanes make no requests), the < 8 trailing lines sit
// one per lane and reach lane 0 through DPP row shifts, the plan's entries are requested before anything else, the sums
// run over registers in NumPy's order, and the only LDS is the 64 node sums per wave: 0.25 ms.
constexpr int kMsWaves = 4;
constexpr int kMsMaxSteps = 16, kMsRounds = 8;          // leaf <= 128 lines; <= 64 leaves, eight per round
template <int CTRL> __device__ __forceinline__ double dpp_from(double v) {           // lane i reads lane i + (CTRL - 0x100) of its row
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
__global__ __launch_bounds__(kWave * kMsWaves) void ms_switch_direct_kernel(int64_t nBlocks, int nBands, int nLeaves,
                                                                             int nInternal, const int* __restrict__ plan,
                                                                             const double* __restrict__ L,
                                                                             const double* __restrict__ R, int64_t blockStride,
                                                                             int* __restrict__ out) {
    __shared__ double sumD[kMsWaves][64], sumS[kMsWaves][64];
    const int lane = threadIdx.x & (kWave - 1), g = lane >> 3, j = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t blk = (int64_t)blockIdx.x * kMsWaves + wave;
    if (blk >= nBlocks) return;                         // (wave-uniform; no workgroup barrier below)
    const double* l = L + blk * blockStride;
    const double* r = R + blk * blockStride;
    // everything the plan says is requested up front: this lane's leaf of every round, the internal node of lane q
    // (children), the root of band `lane`
    int lo[kMsRounds], cnt[kMsRounds];
#pragma unroll
    for (int q = 0; q < kMsRounds; ++q) {
        const int t = 8 * q + g;
        const bool have = t < nLeaves;
        lo[q] = have ? plan[2 * t] : 0;
        cnt[q] = have ? plan[2 * t + 1] : 0;
    }
    const int* inner = plan + 2 * nLeaves;
    const int childA = lane < nInternal ? inner[2 * lane] : 0, childB = lane < nInternal ? inner[2 * lane + 1] : 0;
    const int root = lane < nBands ? plan[2 * nLeaves + 2 * nInternal + lane] : 0;
    auto dOf = [](double a, double b) { return fabs(a * a - b * b); };
    auto sOf = [](double a, double b) { return fabs(a * a + b * b); };
#pragma unroll
    for (int q = 0; q < kMsRounds; ++q) {
        if (8 * q >= nLeaves) break;                    // wave-uniform
        const int n = cnt[q];
        const int steps = n >> 3, body = n & ~7, tail = n & 7;
        double lv[kMsMaxSteps], rv[kMsMaxSteps];
#pragma unroll
        for (int i = 0; i < kMsMaxSteps; ++i) {
            lv[i] = 0.0; rv[i] = 0.0;
            if (i < steps) { lv[i] = l[lo[q] + 8 * i + j]; rv[i] = r[lo[q] + 8 * i + j]; }
        }
        // the < 8 trailing lines of the leaf: lane j holds line body + j
        double lt = 0.0, rt = 0.0;
        if (j < tail) { lt = l[lo[q] + body + j]; rt = r[lo[q] + body + j]; }
        // NumPy's eight strided accumulators of the leaf (lane j holds r[j]), combined in its fixed order; a leaf of fewer
        // than eight lines has none (its sum starts from the 0.0 of np.sum's plain loop: 0.0 + x is x for x >= +0)
        double d = dOf(lv[0], rv[0]), sg = sOf(lv[0], rv[0]);
#pragma unroll
        for (int i = 1; i < kMsMaxSteps; ++i)
            if (i < steps) { d += dOf(lv[i], rv[i]); sg += sOf(lv[i], rv[i]); }
        d += dpp_move<0xB1>(d);   sg += dpp_move<0xB1>(sg);       // r0+r1, r2+r3, ...   (quad_perm [1,0,3,2])
        d += dpp_move<0x4E>(d);   sg += dpp_move<0x4E>(sg);       // (r0+r1)+(r2+r3), ... (quad_perm [2,3,0,1])
        d += dpp_move<0x141>(d);  sg += dpp_move<0x141>(sg);      // + the other quad     (row_half_mirror)
        // ... then the trailing lines one by one, in order: lane 0 of the leaf's eight takes them from its neighbours
        const double dt = dOf(lt, rt), st = sOf(lt, rt);
#define MRC_MS_TAIL(T)                                                                  \
        {                                                                               \
            const double a = T ? dpp_from<0x100 + (T ? T : 1)>(dt) : dt, b = T ? dpp_from<0x100 + (T ? T : 1)>(st) : st;   \
            if (T < tail) { d += a; sg += b; }                                          \
        }
        MRC_MS_TAIL(0) MRC_MS_TAIL(1) MRC_MS_TAIL(2) MRC_MS_TAIL(3) MRC_MS_TAIL(4) MRC_MS_TAIL(5) MRC_MS_TAIL(6)
#undef MRC_MS_TAIL
        if (j == 0 && 8 * q + g < nLeaves) { sumD[wave][8 * q + g] = d; sumS[wave][8 * q + g] = sg; }
    }
    wave_sync();
    // internal nodes (bands of more than 128 lines) in tree order: children before parents, so node q waits for q - 1
    for (int q = 0; q < nInternal; ++q) {
        if (lane == q) {
            sumD[wave][nLeaves + q] = sumD[wave][childA] + sumD[wave][childB];
            sumS[wave][nLeaves + q] = sumS[wave][childA] + sumS[wave][childB];
        }
        wave_sync();
    }
    if (lane < nBands) out[blk * nBands + lane] = sumD[wave][root] < 0.8 * sumS[wave][root] ? 1 : 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// workgroups of mdct_wave_kernel that the current device holds at once: what the LDS of a CU (160 KiB) admits, at most the
// three waves per SIMD its instantiations' registers allow (launch bounds / the compiler's report), times the compute units
static int mdct_wave_slots(size_t lds) {
    static std::mutex mu;
    static std::map<int, int> cusOf;                    // device -> compute units
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cusOf.find(dev);
        if (it != cusOf.end()) cus = it->second;
        else {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            cusOf[dev] = cus;
        }
    }
    const size_t perWg = lds + sizeof(long long) * kMdctWaves * kMdctMaxRun * 4 + 256;   // (+ the static part, rounded up)
    return (int)std::min<size_t>(3, std::max<size_t>(1, (160u << 10) / perWg)) * cus;
}

hipError_t launch_mdct(const DevShape& S, int64_t nFrames, const void* chL, const void* chR, int fmt, int64_t stride,
                       const int64_t* offsets, bool applyWindow, double* lines, int* oscale, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    if (applyWindow && !(reinterpret_cast<uintptr_t>(lines) & 15) && mdct_long_applicable(S, stride, offsets, chL, chR, fmt))
        return launch_mdct_long(S, nFrames, chL, chR, fmt, stride, offsets, lines, oscale, st);
    const int nsig = chR ? 4 : 1;
    // the other shapes of the reference's block switching (short 128 + 128: four units per wavefront at a time; transitions
    // 1024 + 128 and 128 + 1024: one), windowed: mdct_wave_kernel.  Its wavefronts walk runs of up to kMdctMaxRun groups; the
    // number of wavefronts is a whole multiple of what the chip holds at once (mdct_wave_slots), so that a launch of one to
    // three rounds of workgroups -- the block-switched step's sizes -- has no thin last round.
    if (applyWindow && !(reinterpret_cast<uintptr_t>(lines) & 15) && ((S.N == 256 && S.shift == 0) || S.N == 1152)) {
        const int64_t nUnits = nFrames * nsig;
        const int U = S.N == 256 ? 4 : 1;
        const int64_t groups = (nUnits + U - 1) / U;
        // transition blocks fold in registers (shift -+224: chunk pairing 15 / 1)
        const int k64 = S.N == 1152 ? (S.shift == -224 ? 15 : S.shift == 224 ? 1 : -1) : 0;
        if (k64 < 0) return hipErrorInvalidValue;       // (N = 1152 comes from 1024 + 128 or 128 + 1024 only)
        const size_t ldsW = (size_t)(kMdctWaves * (k64 ? S.N : U * S.N + 2 * U * S.Q) + (k64 ? 6 : 2) * S.Q) * sizeof(double);
        int64_t nWaves = 0;
        unsigned grid = 0;
        auto size_launch = [&]() {
            const int64_t perRound = (int64_t)mdct_wave_slots(ldsW) * kMdctWaves;            // wavefronts resident at once
            if (groups <= perRound) nWaves = groups;                                       // one group per wavefront
            else nWaves = (groups + perRound * kMdctMaxRun - 1) / (perRound * kMdctMaxRun) * perRound;
            grid = (unsigned)((nWaves + kMdctWaves - 1) / kMdctWaves);
        };
#define MRC_MDCT_WAVE(TY, NN, UU, NS, KK)                                                                              \
    size_launch();                                                                                                     \
    hipLaunchKernelGGL((mdct_wave_kernel<TY, NN, UU, NS, NN == 256, KK>), dim3(grid), dim3(kWave * kMdctWaves), ldsW, st, S, nUnits, \
                       nWaves, (const TY*)chL, (const TY*)chR, stride, offsets, lines, oscale)
#define MRC_MDCT_WAVE_T(TY)                                                                                            \
    do {                                                                                                               \
        if (S.N == 256) { if (nsig == 1) { MRC_MDCT_WAVE(TY, 256, 4, 1, 0); } else { MRC_MDCT_WAVE(TY, 256, 4, 4, 0); } } \
        else if (k64 == 15) { if (nsig == 1) { MRC_MDCT_WAVE(TY, 1152, 1, 1, 15); } else { MRC_MDCT_WAVE(TY, 1152, 1, 4, 15); } } \
        else { if (nsig == 1) { MRC_MDCT_WAVE(TY, 1152, 1, 1, 1); } else { MRC_MDCT_WAVE(TY, 1152, 1, 4, 1); } }       \
    } while (0)
        if (fmt == kSampleI16) MRC_MDCT_WAVE_T(short); else MRC_MDCT_WAVE_T(double);
#undef MRC_MDCT_WAVE_T
#undef MRC_MDCT_WAVE
        return hipGetLastError();
    }
    size_t lds = (size_t)(2 * S.N) * sizeof(double);
    // a short block (N <= 256: 64 complex FFT points) is one wavefront's work; longer ones take four
#define MRC_MDCT_LAUNCH(TY, THREADS)                                                                                  \
    hipLaunchKernelGGL((mdct_kernel<TY, THREADS>), dim3((unsigned)(nFrames * nsig)), dim3(THREADS), lds, st, S, nsig,   \
                       (const TY*)chL, (const TY*)chR, stride, offsets, applyWindow ? S.win : nullptr, lines, oscale)
    if (fmt == kSampleI16) { if (S.N <= 256) MRC_MDCT_LAUNCH(short, 64); else MRC_MDCT_LAUNCH(short, 256); }
    else { if (S.N <= 256) MRC_MDCT_LAUNCH(double, 64); else MRC_MDCT_LAUNCH(double, 256); }
#undef MRC_MDCT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_quantize_uniform(int64_t n, int nBits, const double* x, long long* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, nBits, x, out);
    return hipGetLastError();
}

hipError_t launch_bark(int64_t n, const double* f, double* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, f, out);
    return hipGetLastError();
}

hipError_t launch_pcm_to_float(int64_t n, const short* pcm, double* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pcm_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, pcm, out);
    return hipGetLastError();
}

hipError_t launch_window(const DevShape& S, int64_t nBlocks, const double* in, double* out, hipStream_t st) {
    const int64_t total = nBlocks * S.N;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(window_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S.N, total, S.win, in,
                       out);
    return hipGetLastError();
}

hipError_t launch_unscale(int64_t nBlocks, int halfN, const double* scaled, const int* oscale, double* lines,
                          hipStream_t st) {
    const int64_t total = nBlocks * halfN;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(unscale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, halfN, total, scaled,
                       oscale, lines);
    return hipGetLastError();
}

hipError_t launch_scale_factor(int64_t n, int nScaleBits, const double* v, const int* nMantBits, int* out,
                               hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_factor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, nScaleBits, v,
                       nMantBits, out);
    return hipGetLastError();
}

hipError_t launch_mantissa(int64_t n, int nScaleBits, const double* x, const int* scale, const int* nMantBits, int* out,
                           hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mantissa_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, nScaleBits, x, scale,
                       nMantBits, out);
    return hipGetLastError();
}

hipError_t launch_transient_peaks(int64_t nHops, int nCh, int hop, int nShort, int nSec, const double* sos, bool unitB0,
                                  const void* streams, int fmt, int64_t chStride, double* peaks, hipStream_t st) {
    const int64_t n = nHops * nCh;
    if (n <= 0) return hipSuccess;
    const size_t sz = fmt == kSampleI16 ? sizeof(short) : sizeof(double);
    const bool vec = !(reinterpret_cast<uintptr_t>(streams) & 15) && !((chStride * sz) & 15) && !((hop * sz) & 15) &&
                     !((nShort * sz) & 15) && nShort > 0 && hop % nShort == 0;
    const dim3 grid((unsigned)((n + kWave - 1) / kWave)), block(kWave);
#define MRC_TP_LAUNCH(TY, NS, VC, UN)                                                                                 \
    hipLaunchKernelGGL((transient_peaks_kernel<TY, NS, VC, UN>), grid, block, 0, st, nHops, nCh, hop, nShort, nSec, sos,  \
                       (const TY*)streams, chStride, peaks)
    // the reference's filter (cheby2 of order 20 through tf2sos: ten sections, unit b0 behind the first) on aligned hops
    // takes the specialised form
    const bool special = nSec == 10 && vec && unitB0;
    if (fmt == kSampleI16) { if (special) MRC_TP_LAUNCH(short, 10, true, true); else MRC_TP_LAUNCH(short, 0, false, false); }
    else { if (special) MRC_TP_LAUNCH(double, 10, true, true); else MRC_TP_LAUNCH(double, 0, false, false); }
#undef MRC_TP_LAUNCH
    return hipGetLastError();
}

hipError_t launch_stereo_masking(int64_t n, const double* mid, const double* side, const double* z, double* outMid,
                                 double* outSide, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(stereo_masking_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, mid, side, z,
                       outMid, outSide);
    return hipGetLastError();
}

hipError_t launch_ms_switch(int64_t nBlocks, int nBands, int nLeaves, int nInternal, const int* plan, const double* L,
                            const double* R, int64_t blockStride, int nLines, int* out, hipStream_t st) {
    if (nBlocks <= 0) return hipSuccess;
    (void)nLines;
    if (nLeaves > 8 * kMsRounds) return hipErrorInvalidValue;      // (build_shape admits at most 64 nodes)
    hipLaunchKernelGGL(ms_switch_direct_kernel, dim3((unsigned)((nBlocks + kMsWaves - 1) / kMsWaves)), dim3(kWave * kMsWaves),
                       0, st, nBlocks, nBands, nLeaves, nInternal, plan, L, R, blockStride, out);
    return hipGetLastError();
}

}  // namespace mrc
