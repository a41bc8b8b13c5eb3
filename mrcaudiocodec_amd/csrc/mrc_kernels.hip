// gfx950 kernels of the encode hot path, generic in the block shape (a,b):
//   mdct_kernel        window.py:104-121 + mdct.py:63-76 + codecThem.py:321-322
//   (smr_kernel lives in mrc_kernels_smr.hip)
//   alloc_quant_kernel ms_stereo.py:5-27,70-81 + bitalloc.py:106-155 + quantize.py:114-146,294-322
//                      + codecThem.py:329-350 / 485-559
// All arithmetic is binary64.  The file is compiled with -ffp-contract=off: wherever the reference's
// operation order decides an integer result (quantiser, bit allocation, SMR) the same separate
// multiplies/adds are issued; fma() is used only where written explicitly.
//
// This is the shape-generic path (any a,b whose N/4 and N/2 factor into 2s and 3s).  Faster
// specialisations for the long block live in mrc_kernels_long.hip.
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
namespace {

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
