// Decode side on gfx950 ("next" row f-4): codecThem.py:30-134 (Decode / JointDecode) with the overlap-and-add of
// pacfileThem.py:312-315 fused, and the 16-bit PCM codes of pcmfile.py:163-172.
//
//   decode_kernel   one workgroup per (block, output channel), any block shape:
//     vDequantize (quantize.py:325-357, operation order kept: ((sign*mag)*2) / (2^R - 1), one correctly rounded
//     division) -> divide by the overall scale level (a power of two: exact) -> ReconstructLR (ms_stereo.py:33-49;
//     a joint block's output channel dequantises BOTH streams of the bands whose M/S switch is set)
//     -> IMDCT (mdct.py:98-122) as a DCT-IV through the same N/4-point complex FFT, twiddles and signed circular
//     shift as the forward kernel (the transform matrix is the transpose of the forward one, so the fold becomes
//     an unfold) -> transition window (window.py:104-121) -> atomic add into the output stream at the block's
//     offset.  Every output sample receives at most two contributions and a + b == b + a, so the result does not
//     depend on the order in which blocks finish.
//   pcm16_kernel    |x| -> 16-bit magnitude code (quantize.py:61-87) with the sign re-applied as 2's complement.
//
// The reference computes the IMDCT with an N-point complex inverse FFT; like the forward transform the results
// agree to ~1e-13 of the block's peak (tests hold 1e-12).
#include "mrc_device.hpp"

namespace mrc {
using namespace dev;
namespace {

// quantize.py:325-357 + 90-111 for one mantissa code
__device__ __forceinline__ double dequantize_dev(int scale, int mant, int nScaleBits, int nMantBits) {
    const int cap = (1 << nScaleBits) - 1;
    const int nBits = cap + nMantBits;
    const int signBit = 1 << (nMantBits - 1);
    const bool neg = mant >= signBit;
    const long long mag = neg ? mant - signBit : mant;
    long long code = mag;
    if (scale != cap) {
        const int shift = cap - scale;
        code = mag << shift;
        if (shift > 0 && mag > 0) code += 1LL << (shift - 1);
    }
    const double sgn = neg ? -1.0 : 1.0;
    return ((sgn * (double)code) * 2.0) / ((double)(1LL << nBits) - 1.0);
}

__global__ __launch_bounds__(kThreads) void decode_kernel(DevShape S, int nStreams, const int* __restrict__ oscale,
                                                          const int* __restrict__ msSwitch,
                                                          const int* __restrict__ scaleFactor,
                                                          const int* __restrict__ bitAlloc,
                                                          const int* __restrict__ mantissa,
                                                          const int64_t* __restrict__ outOffset,
                                                          double* __restrict__ outL, double* __restrict__ outR) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int N = S.N, M = S.halfN, Q = S.Q, nb = S.nBands;
    const int64_t f = blockIdx.x / nStreams;
    const int ch = blockIdx.x % nStreams;
    double* v = smem;                                   // [M] lines, later the DCT-IV output; [M, N) unused
    double2* A = (double2*)(smem + N);                  // [Q]
    double2* B = A + Q;                                 // [Q]
    const bool joint = nStreams == 2;
    const int* sf = scaleFactor + f * nStreams * nb;
    const int* ba = bitAlloc + f * nStreams * nb;
    const int* mant = mantissa + f * nStreams * (int64_t)M;
    const int* os = oscale + f * (joint ? 4 : 1);

    // dequantise, undo the overall scale (codecThem.py:47-51, 92-109), rebuild L / R (ms_stereo.py:33-49)
    for (int k = tid; k < M; k += kThreads) {
        const int band = S.bandOfLine[k];
        double x;
        if (!joint) {
            const int bits = ba[band];
            x = bits ? dequantize_dev(sf[band], mant[k], S.nScaleBits, bits) : 0.0;
            x = ldexp(x, -os[0]);
        } else {
            const bool ms = msSwitch[f * nb + band] == 1;
            const int b0 = ba[band], b1 = ba[nb + band];
            double l1 = b0 ? ldexp(dequantize_dev(sf[band], mant[k], S.nScaleBits, b0), -(ms ? os[2] : os[0])) : 0.0;
            double l2 = b1 ? ldexp(dequantize_dev(sf[nb + band], mant[M + k], S.nScaleBits, b1), -(ms ? os[3] : os[1])) : 0.0;
            x = ms ? (ch == 0 ? l1 + l2 : l1 - l2) : (ch == 0 ? l1 : l2);
        }
        v[k] = x;
    }
    __syncthreads();
    // DCT-IV of the lines through the N/4-point FFT (same pairing and twiddles as the forward kernel)
    for (int n = tid; n < Q; n += kThreads) A[n] = cmul(make_double2(v[2 * n], v[M - 1 - 2 * n]), S.pre[n]);
    __syncthreads();
    double2* T = fft_lds(A, B, Q, S.radQ, S.nRadQ, S.wQ, tid);
    for (int k = tid; k < Q; k += kThreads) {
        const double2 c = cmul(T[k], S.post[k]);
        v[2 * k] = c.x;
        v[M - 1 - 2 * k] = -c.y;
    }
    __syncthreads();
    // unfold M -> N (transpose of the forward fold), undo the signed circular shift, x = 2 y (mdct.py:121 scales by N
    // what its normalised inverse FFT divided by N; the remaining factor is the 2 of the definition), window, add
    double* out = (ch == 0 ? outL : outR) + outOffset[f];
    const int h = Q;
    for (int i = tid; i < N; i += kThreads) {
        int m = i + S.shift;
        double sgn = 2.0;
        if (m < 0) { m += N; sgn = -2.0; }
        else if (m >= N) { m -= N; sgn = -2.0; }
        double y;
        if (m < h) y = v[h + m];
        else if (m < 2 * h) y = -v[3 * h - 1 - m];
        else if (m < 3 * h) y = -v[3 * h - 1 - m];
        else y = -v[m - 3 * h];
        unsafeAtomicAdd(out + i, (sgn * y) * S.win[i]);
    }
}

// pcmfile.py:163-172
__global__ void pcm16_kernel(int64_t n, const double* __restrict__ x, short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double mag = fabs(v);
    const int code = mag == 0.0 ? 0 : (int)mag_code(mag, 16);
    out[i] = (short)(signbit(v) ? -code : code);
}

}  // namespace

hipError_t launch_decode(const DevShape& S, int64_t nBlocks, int nStreams, const int* oscale, const int* msSwitch,
                         const int* scaleFactor, const int* bitAlloc, const int* mantissa, const int64_t* outOffset,
                         double* outL, double* outR, hipStream_t st) {
    if (nBlocks <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * S.N * sizeof(double);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)(nBlocks * nStreams)), dim3(kThreads), lds, st, S, nStreams, oscale,
                       msSwitch, scaleFactor, bitAlloc, mantissa, outOffset, outL, outR);
    return hipGetLastError();
}

hipError_t launch_pcm16(int64_t n, const double* x, short* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x, out);
    return hipGetLastError();
}

}  // namespace mrc
