// Device-side helpers shared by the gfx950 kernels (included by the .hip files only).
#pragma once
#include "mrc_internal.hpp"

namespace mrc {
namespace dev {

constexpr int kThreads = 256;
constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// 16-bit PCM code -> signed fraction exactly as the reference's file reader does (pcmfile.py:91-100 through
// quantize.py:90-111): x = (2 c) / 65535 with ONE rounding (IEEE division), and -32768 -> +0.0 (its magnitude 2^15 is
// read as a bare sign bit).  2 / 65535 = 2^-15 (1 + 2^-16 + 2^-32 + ...) repeats every 16 bits, so its double-double
// form is the same mantissa twice, 64 binades apart, and q = fma(c, kHi, c kLo) is the correctly rounded quotient for
// every one of the 65535 codes (checked exhaustively in exact arithmetic: tests/test_abi.py::test_pcm16_map_is_exact;
// on the device against the reference's own values: tests/test_gpu_pcm16.py).  Conversion, product, fma: three
// instructions per sample where the reciprocal-and-correct form of rounds 1-3 took five.
__device__ __forceinline__ double pcm16_to_frac(int c) {
    const double n = (double)(c == -32768 ? 0 : c);
    return fma(n, 0x1.0001000100010p-15, n * 0x1.0001000100010p-79);
}
// one sample of a channel held as float64 signed fractions or as int16 PCM codes
__device__ __forceinline__ double sample_of(const double* __restrict__ p, int64_t i) { return p[i]; }
__device__ __forceinline__ double sample_of(const short* __restrict__ p, int64_t i) { return pcm16_to_frac(p[i]); }
// an (even, odd) sample pair at an EVEN index: one 16-byte / 4-byte load
__device__ __forceinline__ double2 pair_of(const double* __restrict__ p, int64_t i) {
    return *reinterpret_cast<const double2*>(p + i);
}
__device__ __forceinline__ double2 pair_of(const short* __restrict__ p, int64_t i) {
    const int v = *reinterpret_cast<const int*>(p + i);
    return make_double2(pcm16_to_frac((short)(v & 0xffff)), pcm16_to_frac(v >> 16));
}

// The four signals of a joint block: L, R, M=(L+R)/2, S=(L-R)/2 (codecThem.py:363-364).
template <class T>
__device__ __forceinline__ double load_signal(const T* __restrict__ L, const T* __restrict__ R, int64_t i, int sig) {
    if (sig == 0) return sample_of(L, i);
    if (sig == 1) return sample_of(R, i);
    double l = sample_of(L, i), r = sample_of(R, i);
    return sig == 2 ? (l + r) / 2.0 : (l - r) / 2.0;
}
// ... as (even, odd) pairs; `aligned`: index even and base aligned to the pair size (wave-uniform)
template <class T>
__device__ __forceinline__ double2 load_signal_pair(const T* __restrict__ L, const T* __restrict__ R, int64_t i, int sig,
                                                    bool aligned) {
    if (!aligned)
        return make_double2(load_signal(L, R, i, sig), load_signal(L, R, i + 1, sig));
    if (sig == 0) return pair_of(L, i);
    if (sig == 1) return pair_of(R, i);
    const double2 l = pair_of(L, i), r = pair_of(R, i);
    return sig == 2 ? make_double2((l.x + r.x) / 2.0, (l.y + r.y) / 2.0) : make_double2((l.x - r.x) / 2.0, (l.y - r.y) / 2.0);
}

// MI355X: 8 XCDs, consecutive workgroup ids go to consecutive XCDs.  Maps hardware block id b of a 1-D grid of g
// blocks to a work unit such that XCD x (the blocks with b % 8 == x) gets a contiguous range of units.  A bijection
// for every g.
constexpr unsigned kXcds = 8;
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned g) {
    const unsigned x = b % kXcds, q = b / kXcds;
    const unsigned base = g / kXcds, rem = g % kXcds;
    return x * base + (x < rem ? x : rem) + q;
}

// ---- cross-lane reductions that stay in registers (no ds_bpermute: the LDS pipe is busy enough).  All 64 lanes
// must be active.  Within a 16-lane row: DPP quad permutes, half mirror, rotate by 8.  Across rows: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of one register with the even rows
// (lower half) of another; swapping a value with itself leaves {even, even} and {odd, odd} copies to combine.
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    // every lane has a valid source for the controls used here, so no "old" value has to be prepared
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int DIST, class Op>
__device__ __forceinline__ double rows_combine(double v, Op op) {
    static_assert(DIST == 32 || DIST == 16, "swap distance");
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const uint2v l = DIST == 32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false)
                                : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const uint2v h = DIST == 32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false)
                                : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return op(__hiloint2double((int)h.x, (int)l.x), __hiloint2double((int)h.y, (int)l.y));
}
template <class Op>
__device__ __forceinline__ double wave_allreduce(double v, Op op) {
    v = op(v, dpp_move<0xB1>(v));                       // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E>(v));                       // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141>(v));                      // row_half_mirror: the other quad of the 8-lane group
    v = op(v, dpp_move<0x128>(v));                      // row_ror:8: the other half of the row
    v = rows_combine<16>(v, op);
    return rows_combine<32>(v, op);
}
__device__ __forceinline__ double wave_max(double v) {
    return wave_allreduce(v, [](double a, double b) { return fmax(a, b); });
}
// v_max_f64 as it is: fmax() makes the compiler canonicalise operands that come out of a lane exchange (a second
// v_max_f64 x, x per operand, against signalling NaNs); the values reduced with this are ratios and magnitudes
__device__ __forceinline__ double max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// maximum over each 16-lane row, in every lane of the row
__device__ __forceinline__ double row_max(double v) {
    v = max_raw(v, dpp_move<0xB1>(v));
    v = max_raw(v, dpp_move<0x4E>(v));
    v = max_raw(v, dpp_move<0x141>(v));
    return max_raw(v, dpp_move<0x128>(v));
}
__device__ __forceinline__ double wave_sum(double v) {
    return wave_allreduce(v, [](double a, double b) { return a + b; });
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

template <> __device__ __forceinline__ void butterfly<8>(double2* u) {
    const double h = 0.70710678118654752440;
    auto add = [](double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); };
    auto sub = [](double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); };
    const double2 a0 = add(u[0], u[4]), a1 = add(u[1], u[5]), a2 = add(u[2], u[6]), a3 = add(u[3], u[7]);
    const double2 b0 = sub(u[0], u[4]);
    double2 b1 = sub(u[1], u[5]), b2 = sub(u[2], u[6]), b3 = sub(u[3], u[7]);
    b1 = make_double2(h * (b1.x + b1.y), h * (b1.y - b1.x));          // * (1 - i)/sqrt2
    b2 = make_double2(b2.y, -b2.x);                                   // * (-i)
    b3 = make_double2(h * (b3.y - b3.x), -h * (b3.x + b3.y));         // * (-1 - i)/sqrt2
    double2 e[4] = {a0, a1, a2, a3}, o[4] = {b0, b1, b2, b3};
    butterfly<4>(e);
    butterfly<4>(o);
#pragma unroll
    for (int q = 0; q < 4; ++q) { u[2 * q] = e[q]; u[2 * q + 1] = o[q]; }
}

// Twiddle source of the block FFT.  TwGlobal: the full table exp(-2 pi i t/n) in global memory (any n).
// TwQuarter: its first quadrant staged in LDS, n a power of two: w(s + q n/4) = w(s) (-i)^q -- same values, but no
// global-load latency inside the passes (the FFT of a block is a chain of dependent, barrier-separated passes).
struct TwGlobal {
    const double2* __restrict__ w;
    __device__ __forceinline__ double2 operator()(int t, int /*nq*/) const { return w[t]; }
};
struct TwQuarter {
    const double2* w;                                   // LDS, [n/4]
    int mask, shift;                                    // n/4 - 1, log2(n/4)
    // nq: t is known to lie in quadrants 0 .. nq-1 (a compile-time constant after unrolling, so the unused fix-ups
    // fold away): input r of a radix-R butterfly has t = k r tws < r n/R
    __device__ __forceinline__ double2 operator()(int t, int nq) const {
        const double2 v = w[t & mask];
        if (nq <= 1) return v;
        const int q = t >> shift;
        double2 o = (q & 1) ? make_double2(v.y, -v.x) : v;
        if (nq >= 3 && (q & 2)) o = make_double2(-o.x, -o.y);
        return o;
    }
};

// POW2: p (the product of the radices already done) is a power of two -- true until the first radix-3
// pass, factor() puts the 3s last -- so the index split is a mask and a shift instead of div/mod.
// The first pass (p == 1) has unit twiddles and skips them.
template <int R, bool POW2, class TW, int NT = kThreads>
__device__ __forceinline__ void fft_pass(const double2* __restrict__ in, double2* __restrict__ out, int n, int p,
                                         const TW& W, int tid) {
    const int T = n / R;
    const int tws = n / (p * R);
    const int lp = 31 - __clz(p);
    const bool first = p == 1;
    for (int i = tid; i < T; i += NT) {
        const int k = POW2 ? (i & (p - 1)) : (i % p);
        const int j = (POW2 ? (i >> lp) : (i / p)) * (p * R) + k;
        double2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = in[i + r * T];
        if (!first) {
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul(u[r], W(k * r * tws, (4 * r + R - 1) / R));
        }
        butterfly<R>(u);
#pragma unroll
        for (int q = 0; q < R; ++q) out[j + q * p] = u[q];
    }
}

// Runs all passes; returns the buffer (A or B) that holds the natural-order result.
template <class TW, int NT = kThreads>
__device__ __forceinline__ double2* fft_lds(double2* A, double2* B, int n, const int* rad, int nrad, const TW& W,
                                            int tid) {
    int p = 1;
    for (int s = 0; s < nrad; ++s) {
        const int R = rad[s];
        const bool pow2 = (p & (p - 1)) == 0;
        if (R == 4 && pow2) fft_pass<4, true, TW, NT>(A, B, n, p, W, tid);
        else if (R == 2 && pow2) fft_pass<2, true, TW, NT>(A, B, n, p, W, tid);
        else if (R == 4) fft_pass<4, false, TW, NT>(A, B, n, p, W, tid);
        else if (R == 2) fft_pass<2, false, TW, NT>(A, B, n, p, W, tid);
        else if (pow2) fft_pass<3, true, TW, NT>(A, B, n, p, W, tid);
        else fft_pass<3, false, TW, NT>(A, B, n, p, W, tid);
        __syncthreads();
        double2* t = A; A = B; B = t;
        p *= R;
    }
    return A;
}
// n a power of two (radix 4 / 2 passes only), twiddles from the LDS quadrant
template <int NT = kThreads>
__device__ __forceinline__ double2* fft_lds_pow2(double2* A, double2* B, int n, const int* rad, int nrad,
                                                 const TwQuarter& W, int tid) {
    int p = 1;
    for (int s = 0; s < nrad; ++s) {
        const int R = rad[s];
        if (R == 4) fft_pass<4, true, TwQuarter, NT>(A, B, n, p, W, tid);
        else fft_pass<2, true, TwQuarter, NT>(A, B, n, p, W, tid);
        __syncthreads();
        double2* t = A; A = B; B = t;
        p *= R;
    }
    return A;
}
// The 1024-point transform of the long block (five radix-4 passes) with every stride a compile-time constant: the index
// splits, twiddle strides and quadrant fix-ups fold into immediates.  Same passes, same twiddles, same order as
// fft_lds_pow2 -- bit-identical results.
template <int NT = kThreads>
__device__ __forceinline__ double2* fft_lds_1024(double2* A, double2* B, const double2* wq, int tid) {
    const TwQuarter W{wq, 255, 8};
    fft_pass<4, true, TwQuarter, NT>(A, B, 1024, 1, W, tid);
    __syncthreads();
    fft_pass<4, true, TwQuarter, NT>(B, A, 1024, 4, W, tid);
    __syncthreads();
    fft_pass<4, true, TwQuarter, NT>(A, B, 1024, 16, W, tid);
    __syncthreads();
    fft_pass<4, true, TwQuarter, NT>(B, A, 1024, 64, W, tid);
    __syncthreads();
    fft_pass<4, true, TwQuarter, NT>(A, B, 1024, 256, W, tid);
    __syncthreads();
    return B;
}
// The long block's transform for a thread that already HOLDS the four inputs of its first butterfly (smr_kernel: the windowed
// samples it has just loaded) -- the first pass runs on registers, no trip through the first buffer -- and with the twiddles of
// the other four passes read from a per-pass table in global memory (DevShape::fftTw: for pass p the three factors of
// butterfly k = tid mod p side by side, 48 bytes per lane, requested a pass ahead).  The LDS quadrant cost a gather per factor
// at strides of 64 r, 16 r, 4 r entries -- up to sixteen lanes on one bank -- plus the index and quadrant fix-up arithmetic.
// Same values (the table is built from the quadrant with the same fix-ups), same operations, same order as fft_lds_1024.
struct Tw3 { double2 w[3]; };
__device__ __forceinline__ Tw3 fft1024_twiddles(const double2* __restrict__ tw, int pass /* 1..4 */, int tid) {
    // passes p = 4, 16, 64, 256: tables of 3 p entries behind each other
    const int p = 1 << (2 * pass), base = 3 * ((p - 4) / 3);              // 3 (4 + 16 + ... ) entries before this pass
    const double2* src = tw + base + 3 * (tid & (p - 1));
    return Tw3{{src[0], src[1], src[2]}};
}
// The outputs of the first two passes are written at strides of 4 and 16 entries (of 16 bytes: eight entries span the 32
// banks): without a twist, eight neighbouring lanes would fall on two / four bank groups.  MRC_FFT_TWIST moves entry j of
// the first pass's output to j ^ ((j >> 3) & 3) and entry j of the second's to j ^ (((j >> 4) & 1) << 2): the writes of eight
// lanes then cover all banks, and the next pass's reads of eight CONSECUTIVE entries see a permutation of those eight.
#ifndef MRC_FFT_TWIST
#define MRC_FFT_TWIST 1
#endif
template <int P>
__device__ __forceinline__ void fft1024_pass(const double2* __restrict__ in, double2* __restrict__ out, const Tw3& W, int tid) {
    constexpr int lp = P == 4 ? 2 : P == 16 ? 4 : P == 64 ? 6 : 8;
    const int k = tid & (P - 1);
    const int j = (tid >> lp) * (P * 4) + k;
    int from = tid;
    if (MRC_FFT_TWIST && P == 4) from = tid ^ ((tid >> 3) & 3);
    if (MRC_FFT_TWIST && P == 16) from = tid ^ (((tid >> 4) & 1) << 2);
    const int flip = (MRC_FFT_TWIST && P == 4) ? (tid >> 2) & 1 : 0;
    double2 u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = in[from + r * 256];
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cmul(u[r], W.w[r - 1]);
    butterfly<4>(u);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[j + (q ^ flip) * P] = u[q];
}
// u: in[tid + r 256], r = 0..3; 256 threads.  Returns the buffer with the natural-order result (A).
__device__ __forceinline__ double2* fft_regs_1024(double2 (&u)[4], double2* A, double2* B, const double2* __restrict__ tw,
                                                  Tw3 w1, int tid) {
    butterfly<4>(u);                                     // pass p = 1: unit twiddles
    const int twist = MRC_FFT_TWIST ? (tid >> 1) & 3 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) B[4 * tid + (q ^ twist)] = u[q];
    Tw3 w2 = fft1024_twiddles(tw, 2, tid);
    __syncthreads();
    fft1024_pass<4>(B, A, w1, tid);
    Tw3 w3 = fft1024_twiddles(tw, 3, tid);
    __syncthreads();
    fft1024_pass<16>(A, B, w2, tid);
    Tw3 w4 = fft1024_twiddles(tw, 4, tid);
    __syncthreads();
    fft1024_pass<64>(B, A, w3, tid);
    __syncthreads();
    fft1024_pass<256>(A, B, w4, tid);
    __syncthreads();
    return B;
}
// ... and the 128-point transform of the short block (4 x 4 x 4 x 2)
template <int NT = kThreads>
__device__ __forceinline__ double2* fft_lds_128(double2* A, double2* B, const double2* wq, int tid) {
    const TwQuarter W{wq, 31, 5};
    fft_pass<4, true, TwQuarter, NT>(A, B, 128, 1, W, tid);
    __syncthreads();
    fft_pass<4, true, TwQuarter, NT>(B, A, 128, 4, W, tid);
    __syncthreads();
    fft_pass<4, true, TwQuarter, NT>(A, B, 128, 16, W, tid);
    __syncthreads();
    fft_pass<2, true, TwQuarter, NT>(B, A, 128, 64, W, tid);
    __syncthreads();
    return A;
}
// ... and the 576-point transform of the transition blocks (4 x 4 x 4 x 3 x 3, twiddles from global memory)
template <int NT = kThreads>
__device__ __forceinline__ double2* fft_lds_576(double2* A, double2* B, const double2* __restrict__ w, int tid) {
    const TwGlobal W{w};
    fft_pass<4, true, TwGlobal, NT>(A, B, 576, 1, W, tid);
    __syncthreads();
    fft_pass<4, true, TwGlobal, NT>(B, A, 576, 4, W, tid);
    __syncthreads();
    fft_pass<4, true, TwGlobal, NT>(A, B, 576, 16, W, tid);
    __syncthreads();
    fft_pass<3, true, TwGlobal, NT>(B, A, 576, 64, W, tid);
    __syncthreads();
    fft_pass<3, false, TwGlobal, NT>(A, B, 576, 192, W, tid);
    __syncthreads();
    return B;
}
template <int NT = kThreads>
__device__ __forceinline__ double2* fft_lds_global(double2* A, double2* B, int n, const int* rad, int nrad,
                                                   const double2* __restrict__ W, int tid) {
    return fft_lds<TwGlobal, NT>(A, B, n, rad, nrad, TwGlobal{W}, tid);
}
__device__ __forceinline__ double2* fft_lds(double2* A, double2* B, int n, const int* rad, int nrad,
                                            const double2* __restrict__ W, int tid) {
    return fft_lds_global<kThreads>(A, B, n, rad, nrad, W, tid);
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

}  // namespace dev
}  // namespace mrc
