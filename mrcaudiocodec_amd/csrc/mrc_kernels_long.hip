// Long-block (a = b = 1024, N = 2048) specialisation of the windowed MDCT for gfx950.
//
// One 64-lane wavefront owns one (frame, signal) at a time.  The 512-point complex FFT inside the N/4 MDCT
// algorithm is three radix-8 Stockham passes: 8 points per lane in registers, two lane exchanges through
// LDS (the first one padded by one slot per 8 so that the stride-8 writes stay conflict-free), no workgroup
// barrier in the loop -- only wave-local ordering.  Window values of a lane never change from frame to
// frame and live in registers; the W512 twiddle table sits in LDS, shared by the four waves; pre- and
// post-twiddles factor as (lane constant) x W32^r with compile-time W32 powers.
//
// HBM traffic.  Loads are 16 B per lane over 1 KiB contiguous, stores likewise after an LDS transpose of
// the even/odd output interleave.  Two unit orders:
//   REUSE (mono, hop-overlapped stream): a wave walks kRun CONSECUTIVE frames and keeps the raw second half
//     of its block in registers -- it is the first half of the next block -- so every hop is loaded once:
//     8 KiB in + 8 KiB out per frame, the algorithmic minimum (plus one extra hop per run of kRun frames).
//   otherwise (joint stereo, explicit blocks): the four waves of a workgroup take ADJACENT units at the same
//     time, so what they share (the overlapping hop; the L/R samples of the four signals) is still in L2.
//
// Arithmetic: same formulas as the generic mdct_kernel (window.py:104-121, mdct.py:63-76,
// codecThem.py:321-322), float64, file compiled with -ffp-contract=off.
#include "mrc_device.hpp"

namespace mrc {
namespace {

using dev::kWave;
constexpr int kWavesPerBlock = 4;
constexpr int kM = 1024, kQ = 512;                   // N = 2048: N/2 lines, N/4-point FFT
constexpr int kWaveLds = 2048;                       // doubles per wave (16 KiB)
constexpr int kRun = 16;                             // units per wave

using dev::cmul;
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }     // a * (-i)

// a * e^{-2 pi i R/32}, R = 1..7 (compile-time)
template <int R> __device__ __forceinline__ double2 mul_w32(double2 a) {
    constexpr double c[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                             0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                             0.19509032201612826785};
    return cmul(a, make_double2(c[R], -c[8 - R]));                   // sin(2 pi R/32) = cos(2 pi (8-R)/32)
}

// LDS traffic between lanes of ONE wave: order the wave's own DS operations and stop the compiler from
// moving LDS accesses across this point.  No other wave is involved.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// 4-point DFT (forward): v -> natural order
__device__ __forceinline__ void dft4(double2 v0, double2 v1, double2 v2, double2 v3, double2* o0, double2* o1,
                                     double2* o2, double2* o3) {
    double2 s0 = cadd(v0, v2), s1 = csub(v0, v2), s2 = cadd(v1, v3), s3 = mul_mi(csub(v1, v3));
    *o0 = cadd(s0, s2);
    *o1 = cadd(s1, s3);
    *o2 = csub(s0, s2);
    *o3 = csub(s1, s3);
}

// 8-point DFT (forward, e^{-2 pi i/8}), in place, natural order out
__device__ __forceinline__ void dft8(double2* u) {
    const double h = 0.70710678118654752440;
    double2 a0 = cadd(u[0], u[4]), a1 = cadd(u[1], u[5]), a2 = cadd(u[2], u[6]), a3 = cadd(u[3], u[7]);
    double2 b0 = csub(u[0], u[4]), b1 = csub(u[1], u[5]), b2 = csub(u[2], u[6]), b3 = csub(u[3], u[7]);
    b1 = make_double2(h * (b1.x + b1.y), h * (b1.y - b1.x));          // * (1 - i)/sqrt2
    b2 = mul_mi(b2);                                                  // * (-i)
    b3 = make_double2(h * (b3.y - b3.x), -h * (b3.x + b3.y));         // * (-1 - i)/sqrt2
    dft4(a0, a1, a2, a3, &u[0], &u[2], &u[4], &u[6]);
    dft4(b0, b1, b2, b3, &u[1], &u[3], &u[5], &u[7]);
}

// max over the 64 lanes, in registers (DPP inside a 16-lane row, gfx950 permlane swaps across rows; same scheme as
// dev::wave_allreduce in mrc_device.hpp, which this self-contained file does not include); all lanes active
template <int CTRL> __device__ __forceinline__ double dpp_move(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_move<0xB1>(v));
    v = fmax(v, dpp_move<0x4E>(v));
    v = fmax(v, dpp_move<0x141>(v));
    v = fmax(v, dpp_move<0x128>(v));
    {
        const uint2v l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const uint2v h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = fmax(__hiloint2double((int)h.x, (int)l.x), __hiloint2double((int)h.y, (int)l.y));
    }
    const uint2v l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const uint2v h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return fmax(__hiloint2double((int)h.x, (int)l.x), __hiloint2double((int)h.y, (int)l.y));
}

__device__ __forceinline__ int scale_factor20(double v, int nScaleBits) {        // quantize.py:114-146, nMantBits = 5
    const int cap = (1 << nScaleBits) - 1;
    const int nBits = cap + 5;
    double mag = fabs(v);
    long long code = mag >= 1.0 ? (1LL << (nBits - 1)) - 1
                                : (long long)((((double)((1LL << nBits) - 1)) * mag + 1.0) / 2.0);
    int top = code > 0 ? 63 - __clzll(code) : 0;
    int lz = (nBits - 2) - top;
    return lz < cap ? lz : cap;
}

// one (even, odd) sample pair per lane: 16 bytes of float64 signed fractions or 4 bytes of int16 PCM codes when the
// sample offset is even (always, for strided layouts); an explicit offset may be odd, then two scalar loads
// (wave-uniform choice)
template <int NSIG, class T>
__device__ __forceinline__ void load_pair(const T* __restrict__ L, const T* __restrict__ R, int64_t i, int sig,
                                          double* e, double* o, bool aligned = true) {
    const double2 v = dev::load_signal_pair(L, R, i, NSIG == 1 ? 0 : sig, aligned);     // codecThem.py:363-364 for M, S
    *e = v.x; *o = v.y;
}

// A sample pair as it sits in memory -- 4 bytes of int16 PCM codes or 16 bytes of float64 -- loaded now, converted when
// the unit that needs it starts: the REUSE variants request the NEXT unit's new hop while the current unit is transformed
// (a wave's unit is one dependent chain load -> FFT -> store, and only two waves share a SIMD: without this the HBM
// latency of every hop is exposed).
template <class T> struct RawPair;
template <> struct RawPair<short> {
    int v;
    __device__ __forceinline__ void load(const short* __restrict__ p, int64_t i) { v = *reinterpret_cast<const int*>(p + i); }
    __device__ __forceinline__ double2 get() const { return make_double2(dev::pcm16_to_frac((short)(v & 0xffff)), dev::pcm16_to_frac(v >> 16)); }
};
template <> struct RawPair<double> {
    double2 v;
    __device__ __forceinline__ void load(const double* __restrict__ p, int64_t i) { v = *reinterpret_cast<const double2*>(p + i); }
    __device__ __forceinline__ double2 get() const { return v; }
};
// the pair of signal `sig` (0 L, 1 R, 2 M = (L+R)/2, 3 S = (L-R)/2: codecThem.py:363-364) from the raw pairs of L and R
template <class T>
__device__ __forceinline__ double2 signal_from_raw(const RawPair<T>& l, const RawPair<T>& r, int sig) {
    if (sig == 0) return l.get();
    if (sig == 1) return r.get();
    const double2 a = l.get(), b = r.get();
    return sig == 2 ? make_double2((a.x + b.x) / 2.0, (a.y + b.y) / 2.0) : make_double2((a.x - b.x) / 2.0, (a.y - b.y) / 2.0);
}

// pre[lane + 64 r] = pre[lane] * W32^r and post[lane + 64 q] = post[lane] * W32^q (both tables are unit-circle
// points whose angle is affine in the index with step 2 pi/2048 resp. 4 * 2 pi/4096 per index)
template <int R> __device__ __forceinline__ double2 twiddle_lane(double2 v, double2 laneConst) {
    double2 t = cmul(v, laneConst);
    if (R == 0) return t;
    return mul_w32<R == 0 ? 1 : R>(t);
}

template <int NSIG, bool REUSE, class T>
__global__ __launch_bounds__(kWave * kWavesPerBlock, 2) void mdct_long_kernel(
    DevShape S, int64_t nUnits, const T* __restrict__ chL, const T* __restrict__ chR, int64_t stride,
    const int64_t* __restrict__ offsets, double* __restrict__ lines, int* __restrict__ oscale) {
    __shared__ __attribute__((aligned(16))) double smem[kWavesPerBlock * kWaveLds + 2 * kQ];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // a scalar: unit, signal, offset and every branch on them are wave-uniform
    double* ws = smem + wave * kWaveLds;                               // this wave's 16 KiB
    double2* w512 = reinterpret_cast<double2*>(smem + kWavesPerBlock * kWaveLds);   // [512] e^{-2 pi i t/512}
    for (int i = threadIdx.x; i < kQ; i += kWave * kWavesPerBlock) w512[i] = S.wQ[i];
    // lane-constant registers: window at the lane's sample pairs; pre/post twiddle of the lane's first point
    double wE[16], wO[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double2 w = *reinterpret_cast<const double2*>(S.win + 2 * (lane + 64 * c));
        wE[c] = w.x; wO[c] = w.y;
    }
    const double2 preLane = S.pre[lane];
    const double2 postLane = S.post[lane];
    __syncthreads();                            // W512 table visible to all waves

    // REUSE (hop-overlapped stream): a wave walks kRun CONSECUTIVE FRAMES OF ONE SIGNAL.  Mono: consecutive units.
    // Joint: wave w takes signal w (L, R, M, S) of the workgroup's kRun frames, unit = 4 frame + w, so its units are
    // 4 apart -- and the four waves read the same L / R hops at the same time (one trip from HBM, three L2 hits).
    static_assert(!REUSE || NSIG == 1 || NSIG == kWavesPerBlock, "joint reuse: one wave per signal");
    int64_t firstUnit, step;
    if (REUSE && NSIG == 1) { firstUnit = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kRun; step = 1; }
    else if (REUSE) { firstUnit = (int64_t)blockIdx.x * kRun * NSIG + wave; step = NSIG; }
    else { firstUnit = (int64_t)blockIdx.x * kWavesPerBlock * kRun + wave; step = kWavesPerBlock; }

    double rawE[8], rawO[8];                    // REUSE: raw second half of the previous block (= first half of this one)
    // REUSE: the new hop of the coming unit, requested one unit ahead -- for int16 PCM (8 registers per channel); float64
    // samples would take 32 per channel, more than the wave has left: they are loaded when their unit starts
#ifndef MRC_MDCT_AHEAD                           // 0: never, 1: joint blocks only, 2: mono and joint.  Measured (tools/mdct_bench.py, ms per
                                                 // 131 072 mono / 65 536 joint frames): 0.358 / 1.136, 0.352 / 0.950, 0.361 / 0.945
#define MRC_MDCT_AHEAD 1
#endif
    constexpr bool kAhead = sizeof(T) == 2 && (MRC_MDCT_AHEAD == 2 || (MRC_MDCT_AHEAD == 1 && NSIG != 1));
    RawPair<T> nxtL[kAhead ? 8 : 1], nxtR[kAhead ? 8 : 1];
    int64_t prevOff = 0;
    auto unit_off = [&](int64_t unit) -> int64_t {
        const int64_t f = NSIG == 1 ? unit : unit / NSIG;
        return offsets ? offsets[f] : f * stride;
    };
    auto request_new_hop = [&](int64_t off, int sig) {          // samples [off + 1024, off + 2048): aligned pairs only
        if (!kAhead) return;
#pragma unroll
        for (int c = 0; c < (kAhead ? 8 : 0); ++c) {
            const int64_t i = off + 2 * (lane + 64 * (c + 8));
            if (NSIG == 1 || sig != 1) nxtL[c].load(chL, i);
            if (NSIG != 1 && sig != 0) nxtR[c].load(chR, i);
        }
    };
    if (REUSE && firstUnit < nUnits) {
        const int64_t off = unit_off(firstUnit);
        const int sig0 = (int)(firstUnit % NSIG);
        const bool al = !(off & 1);
#pragma unroll
        for (int c = 0; c < 8; ++c) load_pair<NSIG, T>(chL, chR, off + 2 * (lane + 64 * c), sig0, &rawE[c], &rawO[c], al);
        if (al) request_new_hop(off, sig0);
        prevOff = off - kM;                                      // (so that the first unit counts as a continuation)
    }

    for (int it = 0; it < kRun; ++it) {
        const int64_t unit = firstUnit + (int64_t)it * step;
        if (unit >= nUnits) break;                                     // wave-uniform
        const int64_t f = NSIG == 1 ? unit : unit / NSIG;
        const int sig = NSIG == 1 ? 0 : (int)(unit % NSIG);
        const int64_t off = (REUSE || offsets) ? unit_off(unit) : f * stride;
        const bool aligned = !(off & 1);
        // REUSE: this block continues the previous one of the wave (its first half is in rawE / rawO and its second half
        // was requested a unit ago) -- always in a strided stream, and for explicit offsets whenever they are a hop apart
        // (a block-switched stream is mostly runs of long blocks); else the whole block is loaded here
        const bool cont = REUSE && aligned && off == prevOff + kM;     // wave-uniform

        // ---- A. coalesced load (16 B per lane), window, de-interleave into yE / yO
        double* yE = ws;
        double* yO = ws + kM;
        if (cont) {
            double2 cur[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (kAhead) cur[c] = NSIG == 1 ? nxtL[kAhead ? c : 0].get() : signal_from_raw<T>(nxtL[kAhead ? c : 0], nxtR[kAhead ? c : 0], sig);
                else load_pair<NSIG, T>(chL, chR, off + 2 * (lane + 64 * (c + 8)), sig, &cur[c].x, &cur[c].y, true);
            }
            // the next unit's new hop (if it continues this one; if not, it is loaded when its turn comes)
            const int64_t nu = unit + step;
            if (it + 1 < kRun && nu < nUnits) {
                const int64_t offN = unit_off(nu);
                if (offN == off + kM) request_new_hop(offN, sig);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int i = lane + 64 * c;
                yE[i] = rawE[c] * wE[c];
                yO[i] = rawO[c] * wO[c];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int i = lane + 64 * (c + 8);
                rawE[c] = cur[c].x; rawO[c] = cur[c].y;
                yE[i] = cur[c].x * wE[c + 8];
                yO[i] = cur[c].y * wO[c + 8];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int i = lane + 64 * c;
                double e, o;
                load_pair<NSIG, T>(chL, chR, off + 2 * i, sig, &e, &o, aligned);
                if (REUSE && c >= 8) { rawE[c & 7] = e; rawO[c & 7] = o; }
                yE[i] = e * wE[c];
                yO[i] = o * wO[c];
            }
            if (REUSE) {
                const int64_t nu = unit + step;
                if (it + 1 < kRun && nu < nUnits) {
                    const int64_t offN = unit_off(nu);
                    if (offN == off + kM && !(offN & 1)) request_new_hop(offN, sig);
                }
            }
        }
        prevOff = off;
        wave_sync();
        // ---- B. fold N -> N/2 -> 512 complex points (n = lane + 64 r), pre-twiddle
        double2 u[8];
#define MRC_FOLD(R)                                                                     \
        {                                                                               \
            const int n = lane + 64 * R;                                                \
            double re, im;                                                              \
            if (R < 4) { re = -yO[767 - n] - yE[768 + n]; im = yO[255 - n] - yE[256 + n]; }          \
            else { re = yE[n - 256] - yO[767 - n]; im = -yE[256 + n] - yO[1279 - n]; }  \
            u[R] = twiddle_lane<R>(make_double2(re, im), preLane);                      \
        }
        MRC_FOLD(0) MRC_FOLD(1) MRC_FOLD(2) MRC_FOLD(3) MRC_FOLD(4) MRC_FOLD(5) MRC_FOLD(6) MRC_FOLD(7)
#undef MRC_FOLD
        wave_sync();               // all gathers done before the region is reused
        // ---- C. FFT-512 = 8 x 8 x 8
        dft8(u);
        double2* ex = reinterpret_cast<double2*>(ws);
#pragma unroll
        for (int q = 0; q < 8; ++q) ex[9 * lane + q] = u[q];           // element 8*lane+q, padded (+1 per 8)
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = ex[lane + (lane >> 3) + 72 * r];   // element lane + 64 r
        wave_sync();
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], w512[8 * (lane & 7) * r]);     // W512^(8 k r), k = lane % 8
        dft8(u);
        {
            const int base = 64 * (lane >> 3) + (lane & 7);
#pragma unroll
            for (int q = 0; q < 8; ++q) ex[base + 8 * q] = u[q];
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = ex[lane + 64 * r];
        wave_sync();
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], w512[lane * r]);               // W512^(lane r)
        dft8(u);                                                        // u[q] = T[lane + 64 q]
        // ---- D. post-twiddle, X[2k] = (2/N) Re, X[N/2-1-2k] = -(2/N) Im, transpose through LDS
        double* xE = ws;                                                // xE[i] = X[2i]
        double* xO = ws + kQ;                                           // xO[i] = X[2i+1]
        double peak = 0.0;
#define MRC_POST(Q)                                                                     \
        {                                                                               \
            const int k = lane + 64 * Q;                                                \
            const double2 c = twiddle_lane<Q>(u[Q], postLane);                          \
            const double a = S.twoOverN * c.x;                                          \
            const double b = S.twoOverN * (-c.y);                                       \
            xE[k] = a;                                                                  \
            xO[511 - k] = b;                        /* line 1023-2k = 2*(511-k)+1 */     \
            peak = fmax(peak, fmax(fabs(a), fabs(b)));                                  \
        }
        MRC_POST(0) MRC_POST(1) MRC_POST(2) MRC_POST(3) MRC_POST(4) MRC_POST(5) MRC_POST(6) MRC_POST(7)
#undef MRC_POST
        wave_sync();
        double* dst = lines + unit * kM;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = lane + 64 * c;
            *reinterpret_cast<double2*>(dst + 2 * i) = make_double2(xE[i], xO[i]);
        }
        peak = wave_max(peak);
        if (lane == 0) oscale[unit] = scale_factor20(peak, S.nScaleBits);    // codecThem.py:321-322
        wave_sync();               // output staging read before the next unit overwrites the region
    }
}

}  // namespace

bool mdct_long_applicable(const DevShape& S, int64_t stride, const int64_t* offsets, const void* chL, const void* chR,
                          int fmt) {
    if (S.a != 1024 || S.b != 1024) return false;
    if (!offsets && stride % 2 != 0) return false;   // (explicit offsets: the kernel checks each one's parity itself)
    const uintptr_t mask = fmt == kSampleI16 ? 3 : 15;              // one (even, odd) pair per load
    if ((reinterpret_cast<uintptr_t>(chL) & mask) || (chR && (reinterpret_cast<uintptr_t>(chR) & mask))) return false;
    return true;
}

template <class T>
static void launch_long_t(const DevShape& S, int64_t nFrames, const T* chL, const T* chR, int64_t stride,
                          const int64_t* offsets, double* lines, int* oscale, hipStream_t st) {
    const int nsig = chR ? 4 : 1;
    const int64_t nUnits = nFrames * nsig;
    const int64_t perBlock = (int64_t)kWavesPerBlock * kRun;
    const unsigned grid = (unsigned)((nUnits + perBlock - 1) / perBlock);
    const dim3 block(kWave * kWavesPerBlock);
    // hop-overlapped stream (stride one hop), or explicit offsets (a block-switched stream: mostly runs of blocks a hop
    // apart, detected per block): a wave walks consecutive frames of one signal and keeps / prefetches hops.  Explicit
    // blocks at another stride: the four waves take adjacent units.
    const bool reuse = offsets ? true : stride == kM;
    if (nsig == 4 && reuse)
        hipLaunchKernelGGL((mdct_long_kernel<4, true, T>), dim3(grid), block, 0, st, S, nUnits, chL, chR, stride, offsets,
                           lines, oscale);
    else if (nsig == 4)
        hipLaunchKernelGGL((mdct_long_kernel<4, false, T>), dim3(grid), block, 0, st, S, nUnits, chL, chR, stride, offsets,
                           lines, oscale);
    else if (reuse)
        hipLaunchKernelGGL((mdct_long_kernel<1, true, T>), dim3(grid), block, 0, st, S, nUnits, chL, chR, stride, offsets,
                           lines, oscale);
    else
        hipLaunchKernelGGL((mdct_long_kernel<1, false, T>), dim3(grid), block, 0, st, S, nUnits, chL, chR, stride, offsets,
                           lines, oscale);
}

hipError_t launch_mdct_long(const DevShape& S, int64_t nFrames, const void* chL, const void* chR, int fmt, int64_t stride,
                            const int64_t* offsets, double* lines, int* oscale, hipStream_t st) {
    if (nFrames <= 0) return hipSuccess;
    if (fmt == kSampleI16) launch_long_t(S, nFrames, (const short*)chL, (const short*)chR, stride, offsets, lines, oscale, st);
    else launch_long_t(S, nFrames, (const double*)chL, (const double*)chR, stride, offsets, lines, oscale, st);
    return hipGetLastError();
}

}  // namespace mrc
