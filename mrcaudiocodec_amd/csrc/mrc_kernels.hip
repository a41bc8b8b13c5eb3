// gfx950 kernels of the encode hot path, generic in the block shape (a,b):
//   mdct_kernel        window.py:104-121 + mdct.py:63-76 + codecThem.py:321-322
//   smr_kernel         psychoac.py:134-219 (Hann FFT -> tonal maskers -> masked threshold -> SMR per band)
//   alloc_quant_kernel ms_stereo.py:5-27,70-81 + bitalloc.py:106-155 + quantize.py:114-146,294-322
//                      + codecThem.py:329-350 / 485-559
// All arithmetic is binary64.  The file is compiled with -ffp-contract=off: wherever the reference's
// operation order decides an integer result (quantiser, bit allocation, SMR) the same separate
// multiplies/adds are issued; fma() is used only where written explicitly.
//
// This is the shape-generic path (any a,b whose N/4 and N/2 factor into 2s and 3s).  Faster
// specialisations for the long block live in mrc_kernels_long.hip.
#include "mrc_internal.hpp"

namespace mrc {
namespace {

constexpr int kThreads = 256;
constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// The four signals of a joint block: L, R, M=(L+R)/2, S=(L-R)/2 (codecThem.py:363-364).
__device__ __forceinline__ double load_signal(const double* __restrict__ L, const double* __restrict__ R,
                                               int64_t i, int sig) {
    if (sig == 0) return L[i];
    if (sig == 1) return R[i];
    double l = L[i], r = R[i];
    return sig == 2 ? (l + r) / 2.0 : (l - r) / 2.0;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// psychoac.py:8-12
__device__ __forceinline__ double spl_db(double intensity) {
    return fmax(96 + 10 * log10(intensity), -30.0);
}

// quantize.py:12-38 magnitude code for |x| (R = nBits)
__device__ __forceinline__ long long mag_code(double mag, int nBits) {
    if (mag >= 1.0) return (1LL << (nBits - 1)) - 1;
    return (long long)((((double)((1LL << nBits) - 1)) * mag + 1.0) / 2.0);
}

// quantize.py:114-146
__device__ __forceinline__ int scale_factor_dev(double v, int nScaleBits, int nMantBits) {
    const int cap = (1 << nScaleBits) - 1;
    const int nBits = cap + nMantBits;
    long long code = mag_code(fabs(v), nBits);
    int top = code > 0 ? 63 - __clzll(code) : 0;
    int lz = (nBits - 2) - top;
    return lz < cap ? lz : cap;
}

// quantize.py:294-322 (one element)
__device__ __forceinline__ int mantissa_dev(double x, int scale, int nScaleBits, int nMantBits) {
    const int cap = (1 << nScaleBits) - 1;
    const int nBits = cap + nMantBits;
    long long code = mag_code(fabs(x), nBits);
    int shift = cap - scale;
    if (shift < 0) shift = 0;
    long long m = code >> shift;
    return (int)((x < 0.0 ? (1LL << (nMantBits - 1)) : 0LL) + m);
}

// ------------------------------------------------------------------------------------------------
// mixed-radix Stockham autosort FFT in LDS (forward, e^{-i...}); all threads of the block take part
// ------------------------------------------------------------------------------------------------
template <int R> __device__ __forceinline__ void butterfly(double2* u);

template <> __device__ __forceinline__ void butterfly<2>(double2* u) {
    double2 a = u[0], b = u[1];
    u[0] = make_double2(a.x + b.x, a.y + b.y);
    u[1] = make_double2(a.x - b.x, a.y - b.y);
}

template <> __device__ __forceinline__ void butterfly<3>(double2* u) {
    const double c = 0.86602540378443864676;            // sqrt(3)/2
    double2 t = make_double2(u[1].x + u[2].x, u[1].y + u[2].y);
    double2 d = make_double2(u[1].x - u[2].x, u[1].y - u[2].y);
    double2 m = make_double2(u[0].x - 0.5 * t.x, u[0].y - 0.5 * t.y);
    u[0] = make_double2(u[0].x + t.x, u[0].y + t.y);
    u[1] = make_double2(m.x + c * d.y, m.y - c * d.x);
    u[2] = make_double2(m.x - c * d.y, m.y + c * d.x);
}

template <> __device__ __forceinline__ void butterfly<4>(double2* u) {
    double2 a = make_double2(u[0].x + u[2].x, u[0].y + u[2].y);
    double2 b = make_double2(u[0].x - u[2].x, u[0].y - u[2].y);
    double2 c = make_double2(u[1].x + u[3].x, u[1].y + u[3].y);
    double2 d = make_double2(u[1].x - u[3].x, u[1].y - u[3].y);
    u[0] = make_double2(a.x + c.x, a.y + c.y);
    u[1] = make_double2(b.x + d.y, b.y - d.x);
    u[2] = make_double2(a.x - c.x, a.y - c.y);
    u[3] = make_double2(b.x - d.y, b.y + d.x);
}

template <int R>
__device__ __forceinline__ void fft_pass(const double2* __restrict__ in, double2* __restrict__ out, int n, int p,
                                         const double2* __restrict__ W, int tid) {
    const int T = n / R;
    const int tws = n / (p * R);
    for (int i = tid; i < T; i += kThreads) {
        const int k = i % p;
        const int j = (i / p) * (p * R) + k;
        double2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u[r] = in[i + r * T];
            if (r > 0) u[r] = cmul(u[r], W[k * r * tws]);
        }
        butterfly<R>(u);
#pragma unroll
        for (int q = 0; q < R; ++q) out[j + q * p] = u[q];
    }
}

// Runs all passes; returns the buffer (A or B) that holds the natural-order result.
__device__ double2* fft_lds(double2* A, double2* B, int n, const int* rad, int nrad, const double2* __restrict__ W,
                            int tid) {
    int p = 1;
    for (int s = 0; s < nrad; ++s) {
        const int R = rad[s];
        if (R == 4) fft_pass<4>(A, B, n, p, W, tid);
        else if (R == 2) fft_pass<2>(A, B, n, p, W, tid);
        else fft_pass<3>(A, B, n, p, W, tid);
        __syncthreads();
        double2* t = A; A = B; B = t;
        p *= R;
    }
    return A;
}

// ------------------------------------------------------------------------------------------------
// MDCT: one workgroup per (frame, signal)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void mdct_kernel(DevShape S, int nsig, const double* __restrict__ chL,
                                                        const double* __restrict__ chR, int64_t stride,
                                                        const int64_t* __restrict__ offsets,
                                                        const double* __restrict__ win,
                                                        double* __restrict__ lines, int* __restrict__ oscale) {
    extern __shared__ double smem[];
    __shared__ double red[kThreads / kWave];
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
    for (int n = tid; n < N; n += kThreads) {
        double v = load_signal(chL, chR, off + n, sig);
        if (win) v = v * win[n];
        int m = n + S.shift;
        if (m < 0) { m += N; v = -v; }
        else if (m >= N) { m -= N; v = -v; }
        y[m] = v;
    }
    __syncthreads();
    // fold N -> N/2 (DCT-IV input u) and pack pairs into Q complex points with the pre-twiddle
    for (int n = tid; n < Q; n += kThreads) {
        const int j0 = 2 * n, j1 = M - 1 - 2 * n, h = Q;
        double u0 = (j0 < h) ? (-y[3 * h - 1 - j0] - y[3 * h + j0]) : (y[j0 - h] - y[3 * h - 1 - j0]);
        double u1 = (j1 < h) ? (-y[3 * h - 1 - j1] - y[3 * h + j1]) : (y[j1 - h] - y[3 * h - 1 - j1]);
        A[n] = cmul(make_double2(u0, u1), S.pre[n]);
    }
    __syncthreads();
    double2* T = fft_lds(A, B, Q, S.radQ, S.nRadQ, S.wQ, tid);
    // post-twiddle; y is free again (all reads of it happened before the FFT's first barrier)
    for (int k = tid; k < Q; k += kThreads) {
        double2 c = cmul(T[k], S.post[k]);
        y[2 * k] = S.twoOverN * c.x;
        y[M - 1 - 2 * k] = S.twoOverN * (-c.y);
    }
    __syncthreads();
    double peak = 0.0;
    double* dst = lines + ((int64_t)blockIdx.x) * M;
    for (int k = tid; k < M; k += kThreads) {
        double v = y[k];
        dst[k] = v;
        peak = fmax(peak, fabs(v));
    }
    peak = wave_max(peak);
    if ((tid & (kWave - 1)) == 0) red[tid / kWave] = peak;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kThreads / kWave; ++w) peak = fmax(peak, red[w]);
        oscale[blockIdx.x] = scale_factor_dev(peak, S.nScaleBits, 5);       // codecThem.py:322 (nMantBits default)
    }
}

// ------------------------------------------------------------------------------------------------
// SMR: one workgroup per (frame, signal)
// ------------------------------------------------------------------------------------------------
constexpr int kLinesPerThread = 4;                     // register tile: kThreads*4 = 1024 lines per sweep
constexpr double kLog2Of10 = 3.32192809488736234787;
constexpr double kLowerSlopeBits = -2.7 * 3.32192809488736234787;   // -27 dB/Bark below the masker (psychoac.py:74)

// 2^t for finite t (any sign; used with t <= 0): k = rint(t), 2^(t-k) by a degree-11 polynomial on
// [-0.5, 0.5] (Chebyshev-node fit, max relative error 2e-16 incl. evaluation), scaled by 2^k.
// t == 0 returns exactly 1, so a line inside +-0.5 Bark gets exactly the masker's own intensity.
__device__ __forceinline__ double exp2_neg(double t) {
    const double k = rint(t);
    const double f = t - k;
    double p = 0x1.e9ec1fcb69a7fp-32;
    p = fma(p, f, 0x1.e6228acd1c6e5p-28);
    p = fma(p, f, 0x1.b524ebd13a55fp-24);
    p = fma(p, f, 0x1.62bfc2c86d700p-20);
    p = fma(p, f, 0x1.ffcbfc6da6ed1p-17);
    p = fma(p, f, 0x1.430913112c61bp-13);
    p = fma(p, f, 0x1.5d87fe78a3f9cp-10);
    p = fma(p, f, 0x1.3b2ab6fb9f1a5p-7);
    p = fma(p, f, 0x1.c6b08d704a0c6p-5);
    p = fma(p, f, 0x1.ebfbdff82c5aep-3);
    p = fma(p, f, 0x1.62e42fefa39efp-1);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)k);
}

// EXACT = true keeps the reference's per-(masker, line) expression with pow(); EXACT = false (default)
// evaluates the same quantity factored as I_m * 2^(slope_m * u): ~8x fewer instructions, same integers
// on every parity corpus (tests/test_gpu_parity.py::test_spread_modes_agree).
template <bool EXACT>
__global__ __launch_bounds__(kThreads) void smr_kernel(DevShape S, int nsig, const double* __restrict__ chL,
                                                       const double* __restrict__ chR, int64_t stride,
                                                       const int64_t* __restrict__ offsets,
                                                       const double* __restrict__ lines,
                                                       const int* __restrict__ oscale, double* __restrict__ smr,
                                                       double* __restrict__ thresh) {
    extern __shared__ double smem[];
    __shared__ int cnt[kThreads];
    const int tid = threadIdx.x;
    const int H = S.H, M = S.halfN;
    const int64_t f = blockIdx.x / nsig;
    const int sig = blockIdx.x % nsig;
    const int64_t off = offsets ? offsets[f] : f * stride;
    double2* A = (double2*)smem;                        // [H]
    double2* B = A + H;                                 // [H]
    double* xi = smem + 4 * H;                          // [H] intensity spectrum (bins < peakLast used)

    // Hann window (window.py:28-45) and real FFT through an H = N/2 point complex FFT
    for (int n = tid; n < H; n += kThreads) {
        double e = load_signal(chL, chR, off + 2 * n, sig) * S.hann[2 * n];
        double o = load_signal(chL, chR, off + 2 * n + 1, sig) * S.hann[2 * n + 1];
        A[n] = make_double2(e, o);
    }
    __syncthreads();
    double2* T = fft_lds(A, B, H, S.radH, S.nRadH, S.wH, tid);
    const int last = S.peakLast;                        // bins 0 .. last-1 are inspected (psychoac.py:160)
    for (int k = tid; k < last; k += kThreads) {
        double2 zk = T[k];
        double2 zc = T[(H - k) % H];
        zc.y = -zc.y;
        double2 ev = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
        double2 d = make_double2(zk.x - zc.x, zk.y - zc.y);
        double2 od = make_double2(0.5 * d.y, -0.5 * d.x);
        double2 X = cmul(S.wN[k], od);
        X.x += ev.x; X.y += ev.y;
        xi[k] = 4. * (X.x * X.x + X.y * X.y) / S.xiDen;  // psychoac.py:151
    }
    __syncthreads();                                    // T (in A or B) is dead from here on

    // tonal maskers: strict 3-point peaks at bins p = 1 .. last-2, kept in increasing bin order
    // per masker: [0] level-15 dB (EXACT mode) or its intensity 10^((level-15-96)/10) (fast mode),
    //             [1] Bark position, [2] 0.37*max(level-40,0) (EXACT) or the upper slope in bits per Bark (fast)
    double* mLvl = smem;                                // (aliases A)
    double* mZ = mLvl + H / 2 + 1;
    double* mBoost = mZ + H / 2 + 1;
    const int nCand = last - 2;
    const int per = (nCand + kThreads - 1) / kThreads;
    const int p0 = 1 + tid * per;
    const int p1 = min(p0 + per, last - 1);
    int mine = 0;
    for (int p = p0; p < p1; ++p) mine += (xi[p] > xi[p - 1] && xi[p] > xi[p + 1]) ? 1 : 0;
    cnt[tid] = mine;
    __syncthreads();
    int before = 0, nPeaks = 0;
    for (int t = 0; t < kThreads; ++t) {
        int c = cnt[t];
        before += (t < tid) ? c : 0;
        nPeaks += c;
    }
    for (int p = p0; p < p1; ++p) {
        double x0 = xi[p - 1], x1 = xi[p], x2 = xi[p + 1];
        if (x1 > x0 && x1 > x2) {
            double s3 = (x0 + x1) + x2;
            double level = spl_db(s3);                                       // psychoac.py:164
            double fm = S.binHz * (((p - 1) * x0 + p * x1) + (p + 1) * x2) / s3;   // psychoac.py:165
            double q = fm / 7500.;
            mZ[before] = 13 * atan(0.76 * fm / 1000.) + 3.5 * atan(q * q);   // psychoac.py:27-29
            const double lvl15 = level - 15.0;                               // psychoac.py:42-43 (tonal drop)
            const double boost = 0.37 * fmax(level - 40, 0.0);               // psychoac.py:76
            if (EXACT) {
                mLvl[before] = lvl15;
                mBoost[before] = boost;
            } else {
                mLvl[before] = pow(10.0, (lvl15 - 96) / 10);                 // value inside +-0.5 Bark, psychoac.py:14-18
                mBoost[before] = ((-27 + boost) / 10) * kLog2Of10;           // dB/Bark above the masker -> bits/Bark
            }
            ++before;
        }
    }
    __syncthreads();

    double* excess = smem + 2 * H;                      // [M] (aliases B)
    const int scale = oscale[blockIdx.x];
    const double* X = lines + (int64_t)blockIdx.x * M;
    for (int base = 0; base < M; base += kThreads * kLinesPerThread) {
        double z[kLinesPerThread], tot[kLinesPerThread];
#pragma unroll
        for (int j = 0; j < kLinesPerThread; ++j) {
            int k = base + tid + j * kThreads;
            bool ok = k < M;
            z[j] = ok ? S.zb[k] : 0.0;
            tot[j] = ok ? S.quiet[k] : 0.0;
        }
        // psychoac.py:166-168 + 68-78: add every masker's spread intensity, in masker order
        for (int m = 0; m < nPeaks; ++m) {
            const double lvl = mLvl[m], zm = mZ[m], boost = mBoost[m];
#pragma unroll
            for (int j = 0; j < kLinesPerThread; ++j) {
                double dz = z[j] - zm;
                if (EXACT) {                             // the reference's expression, operation by operation
                    double adz = fabs(dz);
                    double t = adz - 0.5;
                    double arg = lvl;
                    if (adz > 0.5) arg = lvl + (-27 * t);
                    if (dz > 0.5) arg = arg + boost * t;
                    tot[j] += pow(10.0, (arg - 96) / 10);
                } else {                                 // same quantity as I_m * 2^(slope * max(|dz|-0.5, 0))
                    double u = fmax(fabs(dz) - 0.5, 0.0);
                    double slope = dz > 0.0 ? boost : kLowerSlopeBits;
                    tot[j] = fma(lvl, exp2_neg(slope * u), tot[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kLinesPerThread; ++j) {
            int k = base + tid + j * kThreads;
            if (k < M) {
                double thr = spl_db(tot[j]);                                 // psychoac.py:173
                if (thresh) thresh[(int64_t)blockIdx.x * M + k] = thr;
                double xs = ldexp(X[k], scale);                              // codecThem.py:323 (exact)
                double spl = spl_db(2. * (xs * xs) / (1. / 2.)) - 6. * scale;   // psychoac.py:212
                excess[k] = spl - thr;
            }
        }
    }
    __syncthreads();
    // psychoac.py:216-217: SMR of a band = max over its lines
    for (int bnd = tid; bnd < S.nBands; bnd += kThreads) {
        const int lo = S.bandLo[bnd], n = S.bandN[bnd];
        double best = excess[lo];
        for (int k = 1; k < n; ++k) best = fmax(best, excess[lo + k]);
        smr[(int64_t)blockIdx.x * S.nBands + bnd] = best;
    }
}

// ------------------------------------------------------------------------------------------------
// bit allocation (bitalloc.py:106-155) on one wavefront: lane i owns band i
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitalloc_wave(double smr, int nLines, bool active, double budget, int maxMantBits,
                                              int nTot, int lane, int* bitsOut, double* leftOut) {
    double run = active ? smr : -INFINITY;
    int bits = 0;
    double left = budget;
    int retired = 0;
    // every iteration either grants (<= maxMantBits-1 times per band) or retires (<= nTot times):
    // the loop ends by itself; the counter is a guard so that no wave can spin on bad input (NaN).
    int guard = (maxMantBits + 2) * nTot + 8;
    while (left > 0 && guard-- > 0) {
        double v = run;
        int idx = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(v, off);
            int oi = __shfl_xor(idx, off);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }     // np.argmax: first maximum
        }
        const int wb = __shfl(bits, idx);
        const int wn = __shfl(nLines, idx);
        if (wb < maxMantBits && (double)wn <= left) {
            if (wb == 0) {
                if (lane == idx) { bits += 2; run -= 12.0; }
                left -= (double)(2 * wn);
            } else {
                if (lane == idx) { bits += 1; run -= 6.0; }
                left -= (double)wn;
            }
        } else {
            if (lane == idx) run = -99999999999999999.0;
            if (++retired == nTot) break;
        }
    }
    *bitsOut = bits;
    *leftOut = left;
}

// NumPy's pairwise summation of a contiguous run (np.sum over a 1-D slice), restated so that the
// M/S decision sums round exactly like ms_stereo.py:19-20.  T(i) yields element i.
template <class F> __device__ double pairwise_sum(F elem, int lo, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += elem(lo + i);
        return r;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = elem(lo + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += elem(lo + i + j);
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += elem(lo + i);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(elem, lo, n2) + pairwise_sum(elem, lo + n2, n - n2);
}

// ms_stereo.py:5-27 for one band
__device__ __forceinline__ int ms_switch_band(const double* __restrict__ L, const double* __restrict__ R, int lo,
                                              int n) {
    double d = pairwise_sum([&](int k) { double l = L[k], r = R[k]; return fabs(l * l - r * r); }, lo, n);
    double s = pairwise_sum([&](int k) { double l = L[k], r = R[k]; return fabs(l * l + r * r); }, lo, n);
    return d < 0.8 * s ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// M/S decision + SMR select + bit allocation + scale factors + mantissas: one wavefront per frame
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void alloc_quant_kernel(DevShape S, int joint, const double* __restrict__ lines,
                                                            const int* __restrict__ oscale,
                                                            const double* __restrict__ smr,
                                                            const int* __restrict__ resIn, int* __restrict__ msSwitch,
                                                            int* __restrict__ bitAlloc, int* __restrict__ scaleFactor,
                                                            int* __restrict__ mantissa, int* __restrict__ resOut) {
    __shared__ int sSw[kMaxBands];
    __shared__ int sBa[2 * kMaxBands];
    __shared__ int sSf[2 * kMaxBands];
    const int lane = threadIdx.x;
    const int64_t f = blockIdx.x;
    const int M = S.halfN, nb = S.nBands;
    const int nsig = joint ? 4 : 1, nstream = joint ? 2 : 1;
    const double* X = lines + f * nsig * M;
    const int* osc = oscale + f * nsig;

    if (joint) {
        if (lane < nb) {                                 // on the UNSCALED L/R lines (codecThem.py:436)
            int sw = ms_switch_band(X, X + M, S.bandLo[lane], S.bandN[lane]);
            sSw[lane] = sw;
            msSwitch[f * nb + lane] = sw;
        }
    } else if (lane < nb) {
        sSw[lane] = 0;
    }
    __syncthreads();

    const int nTot = nstream * nb;
    const bool active = lane < nTot;
    const int band = active ? lane % nb : 0;
    const int strm = active ? lane / nb : 0;
    // stream 0 = Mid-or-Left, stream 1 = Side-or-Right (ms_stereo.py:70-81, codecThem.py:524-551)
    const int sig = joint ? (sSw[band] ? 2 + strm : strm) : 0;
    const double mySmr = active ? smr[(f * nsig + sig) * nb + band] : 0.0;
    const int myLines = active ? S.bandN[band] : 0;
    const double r = resIn ? (double)resIn[f] : 0.0;
    double budget;
    if (joint) { budget = S.budgetJointPre + r; budget -= S.blkswA; budget -= S.blkswB; }   // codecThem.py:390-396
    else budget = S.budgetMono + r;                                                           // codecThem.py:308
    int bits;
    double left;
    bitalloc_wave(mySmr, myLines, active, budget, S.maxMantBits, nTot, lane, &bits, &left);
    if (lane == 0) resOut[f] = (int)left;               // int(bitsLeft): truncation toward zero (bitalloc.py:155)

    if (active) {
        // codecThem.py:346-347: scale factor from the band's largest |scaled line| with nMantBits = bitAlloc
        const double* Xs = X + sig * M;
        const int lo = S.bandLo[band];
        double peak = 0.0;
        for (int k = 0; k < myLines; ++k) peak = fmax(peak, fabs(Xs[lo + k]));
        peak = ldexp(peak, osc[sig]);
        int sf = scale_factor_dev(peak, S.nScaleBits, bits);
        sBa[lane] = bits;
        sSf[lane] = sf;
        bitAlloc[f * nTot + lane] = bits;
        scaleFactor[f * nTot + lane] = sf;
    }
    __syncthreads();
    for (int s = 0; s < nstream; ++s) {
        int* out = mantissa + (f * nstream + s) * M;
        for (int k = lane; k < M; k += kWave) {
            const int bnd = S.bandOfLine[k];
            const int ba = sBa[s * nb + bnd];
            int m = 0;
            if (ba) {
                const int sg = joint ? (sSw[bnd] ? 2 + s : s) : 0;
                double x = ldexp(X[sg * M + k], osc[sg]);
                m = mantissa_dev(x, sSf[s * nb + bnd], S.nScaleBits, ba);    // codecThem.py:348-349
            }
            out[k] = m;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stage-level kernels for the parity tests against the reference's own modules
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void bitalloc_cases_kernel(int nBands, int maxMantBits,
                                                               const int* __restrict__ nLines,
                                                               const double* __restrict__ budget,
                                                               const double* __restrict__ smr, int* __restrict__ bits,
                                                               int* __restrict__ left) {
    const int lane = threadIdx.x;
    const int64_t c = blockIdx.x;
    const bool active = lane < nBands;
    int b;
    double l;
    bitalloc_wave(active ? smr[c * nBands + lane] : 0.0, active ? nLines[lane] : 0, active, budget[c], maxMantBits,
                  nBands, lane, &b, &l);
    if (active) bits[c * nBands + lane] = b;
    if (lane == 0) left[c] = (int)l;
}

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

__global__ __launch_bounds__(kWave) void ms_switch_kernel(int nBands, int nTotal, const int* __restrict__ bandLo,
                                                          const int* __restrict__ bandN, const double* __restrict__ L,
                                                          const double* __restrict__ R, int* __restrict__ out) {
    const int64_t blk = blockIdx.x;
    for (int b = threadIdx.x; b < nBands; b += kWave)
        out[blk * nBands + b] = ms_switch_band(L + blk * nTotal, R + blk * nTotal, bandLo[b], bandN[b]);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
hipError_t launch_mdct(const DevShape& S, int64_t nFrames, const double* chL, const double* chR, int64_t stride,
                       const int64_t* offsets, bool applyWindow, double* lines, int* oscale, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    if (applyWindow && !(reinterpret_cast<uintptr_t>(lines) & 15) && mdct_long_applicable(S, stride, offsets, chL, chR))
        return launch_mdct_long(S, nFrames, chL, chR, stride, lines, oscale, st);
    const int nsig = chR ? 4 : 1;
    size_t lds = (size_t)(2 * S.N) * sizeof(double);
    hipLaunchKernelGGL(mdct_kernel, dim3((unsigned)(nFrames * nsig)), dim3(kThreads), lds, st, S, nsig, chL, chR,
                       stride, offsets, applyWindow ? S.win : nullptr, lines, oscale);
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

hipError_t launch_smr(const DevShape& S, int64_t nFrames, const double* chL, const double* chR, int64_t stride,
                      const int64_t* offsets, const double* lines, const int* oscale, double* smr, double* thresh,
                      bool exactSpread, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    const int nsig = chR ? 4 : 1;
    size_t lds = (size_t)(5 * S.H) * sizeof(double);
    if (exactSpread)
        hipLaunchKernelGGL(smr_kernel<true>, dim3((unsigned)(nFrames * nsig)), dim3(kThreads), lds, st, S, nsig, chL,
                           chR, stride, offsets, lines, oscale, smr, thresh);
    else
        hipLaunchKernelGGL(smr_kernel<false>, dim3((unsigned)(nFrames * nsig)), dim3(kThreads), lds, st, S, nsig, chL,
                           chR, stride, offsets, lines, oscale, smr, thresh);
    return hipGetLastError();
}

hipError_t launch_alloc_quant(const DevShape& S, int64_t nFrames, int joint, const double* lines, const int* oscale,
                              const double* smr, const int* resIn, int* msSwitch, int* bitAlloc, int* scaleFactor,
                              int* mantissa, int* resOut, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    hipLaunchKernelGGL(alloc_quant_kernel, dim3((unsigned)nFrames), dim3(kWave), 0, st, S, joint, lines, oscale, smr,
                       resIn, msSwitch, bitAlloc, scaleFactor, mantissa, resOut);
    return hipGetLastError();
}

hipError_t launch_bitalloc_cases(int64_t nCases, int nBands, int maxMantBits, const int* nLines, const double* budget,
                                 const double* smr, int* bits, int* left, hipStream_t st) {
    if (nCases <= 0) return hipSuccess;
    hipLaunchKernelGGL(bitalloc_cases_kernel, dim3((unsigned)nCases), dim3(kWave), 0, st, nBands, maxMantBits, nLines,
                       budget, smr, bits, left);
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

hipError_t launch_ms_switch(int64_t nBlocks, int nBands, int nTotal, const int* bandLo, const int* bandN,
                            const double* L, const double* R, int* out, hipStream_t st) {
    if (nBlocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(ms_switch_kernel, dim3((unsigned)nBlocks), dim3(kWave), 0, st, nBands, nTotal, bandLo, bandN, L,
                       R, out);
    return hipGetLastError();
}

}  // namespace mrc
