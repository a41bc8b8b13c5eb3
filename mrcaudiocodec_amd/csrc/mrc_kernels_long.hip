// Long-block (a = b = 1024, N = 2048) specialisation of the windowed MDCT for gfx950.
//
// One 64-lane wavefront owns one (frame, signal) at a time.  The windowed block is folded into the 512 complex points
// of the N/4 MDCT algorithm in the lane's own registers (each lane loads the even samples that run with its index and the
// odd samples that run against it: see mdct_long_kernel); the 512-point FFT is three radix-8 Stockham passes, 8 points per
// lane in registers, two lane exchanges through LDS (the first one padded by one slot per 8 so that the stride-8 writes
// stay conflict-free), no workgroup barrier in the loop -- only wave-local ordering.  Window values of a lane never change
// from frame to frame and live in registers (one half: the long block's window is symmetric); the W512 twiddle table sits
// in LDS, shared by the four waves; pre- and post-twiddles factor as (lane constant) x W32^r with compile-time W32 powers.
// LDS: 9 KiB per wave (the exchange buffer, reused for the output transpose) + 8 KiB of twiddles = 44 KiB per workgroup.
//
// HBM traffic.  int16 PCM is read as 4-byte words, 256 B per wave-load; float64 lines leave as 16 B per lane over 1 KiB
// contiguous after an LDS transpose of the even/odd output interleave.  Two unit orders:
//   REUSE (hop-overlapped stream, or explicit offsets a hop apart): a wave walks CONSECUTIVE frames of one signal and keeps
//     the raw second hop of its block in registers -- it is the first hop of the next block -- and requests the hop after it
//     one unit ahead: every hop is loaded once per signal, 2 KiB in + 8 KiB out per mono frame (plus one extra hop per
//     run).  Joint: wave w of a workgroup walks signal w (L, R, M, S) over half of the workgroup's frames and signal
//     (w + 2) % 4 over the other half (the M / S units convert both channels: this way every wave does as many of each).
//   otherwise (explicit blocks at another stride): the four waves of a workgroup take ADJACENT units at the same
//     time, so what they share (the overlapping hop; the L/R samples of the four signals) is still in L2.
//
// What bounds it (profiles/r04_mdct_stop_variants.txt: the kernel without stores / sample loads / LDS exchanges):
// fp64 issue.  ~700 VALU instructions per unit at 4 cycles each on two waves per SIMD is 0.21 ms per 131 072 mono units
// with no memory operation at all; the LDS exchanges hide completely; the stores add 0.05 ms, the loads the rest.  A third
// workgroup per CU (registers allow it for mono) measured SLOWER, as did 8- or 32-unit runs and non-temporal stores.
//
// Arithmetic: same formulas as the generic mdct_kernel (window.py:104-121, mdct.py:63-76,
// codecThem.py:321-322), float64, file compiled with -ffp-contract=off.
#include <type_traits>

#include "mrc_device.hpp"

namespace mrc {
namespace {

using dev::kWave;
constexpr int kWavesPerBlock = 4;
constexpr int kM = 1024, kQ = 512;                   // N = 2048: N/2 lines, N/4-point FFT
constexpr int kWaveLds = 1152;                       // doubles per wave (9 KiB): the padded exchange buffer of the FFT (576 complex) / the output transpose (1024 lines)
#ifndef MRC_MDCT_WG_PER_CU
#define MRC_MDCT_WG_PER_CU 2
#endif
#ifndef MRC_MDCT_PROFILE                             // timing experiments only (wrong results): 1 no stores, 2 no sample loads, 4 no LDS exchanges
#define MRC_MDCT_PROFILE 0
#endif
constexpr int kRun = 16;                             // units per wave (<= 64: a lane per unit holds its offset; 8 and 32 measured no faster)

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

// One raw sample as it sits in memory -- an int16 PCM code or a float64 signed fraction -- loaded now, converted when
// the unit that needs it starts: the REUSE variants request the NEXT unit's new hop while the current unit is transformed
// (a wave's unit is one dependent chain load -> FFT -> store: without this the HBM latency of every hop is exposed).
// Conversion with a wave-uniform sign: the S signal of a joint block is (L + (-R)) / 2 (codecThem.py:363-364; a - b and
// a + (-b) are the same IEEE operation), and -R comes for free from signed conversion constants -- round-to-nearest is
// symmetric, fma(n, -k1, n * -k2) == -fma(n, k1, n * k2) bit for bit (dev::pcm16_to_frac's two-part 1 / 32767).
struct SampleConv { double k1, k2, sgn; };
__device__ __forceinline__ SampleConv sample_conv(bool negate) {
    const double s = negate ? -1.0 : 1.0;
    return SampleConv{s * 0x1.0001000100010p-15, s * 0x1.0001000100010p-79, s};
}
// Where a lane's samples of one channel sit for the block at sample offset `off`: stretch c (128 samples) holds the lane's
// even sample at 2 lane + 128 c and its odd one at 2 (63 - lane) + 1 + 128 c.  int16 PCM is read as aligned 32-bit words
// (full-rate loads whatever the texture path does with 2-byte ones; measured 1 % faster than the 2-byte form) and the wanted
// half is cut out with one v_bfe_i32 whose bit offset is a scalar: which half it is depends only on the parity of (channel
// base + off), the same for all lanes.  A word read for an odd parity reaches one sample before / after the block, never
// outside the aligned 4-byte word that holds a sample of it.
template <class T> struct ChanView;
template <> struct ChanView<short> {
    typedef int Raw;
    const int* e; const int* o;          // wave-uniform word pointers of stretch 0 (lane index added at the load)
    unsigned shE, shO;                   // bit offset of the even / odd sample inside its word
    __device__ __forceinline__ void set(const short* __restrict__ p, int64_t off) {
        const int64_t mis = (int64_t)(reinterpret_cast<uintptr_t>(p) & 2);                 // base on an odd sample of its word?
        const int* w0 = reinterpret_cast<const int*>(reinterpret_cast<const char*>(p) - mis); // (pointer arithmetic: stays a global pointer)
        const int64_t s0 = off + (mis >> 1);                                                 // the block's first sample, counted from w0
        e = w0 + (s0 >> 1);
        o = w0 + ((s0 + 1) >> 1);
        shE = 16u * (unsigned)(s0 & 1);
        shO = 16u * (unsigned)((s0 + 1) & 1);
    }
    __device__ __forceinline__ Raw loadE(int lane, int c) const { return (MRC_MDCT_PROFILE & 2) ? lane * c : e[lane + 64 * c]; }
    __device__ __forceinline__ Raw loadO(int lane, int c) const { return (MRC_MDCT_PROFILE & 2) ? lane + c : o[(63 - lane) + 64 * c]; }
    static __device__ __forceinline__ double conv(int v, const SampleConv& k) {
        const double n = (double)(v == -32768 ? 0 : v);
        return fma(n, k.k1, n * k.k2);
    }
    __device__ __forceinline__ double getE(Raw w, const SampleConv& k) const { return conv(__builtin_amdgcn_sbfe(w, shE, 16u), k); }
    __device__ __forceinline__ double getO(Raw w, const SampleConv& k) const { return conv(__builtin_amdgcn_sbfe(w, shO, 16u), k); }
};
template <> struct ChanView<double> {
    typedef double Raw;
    const double* b;
    __device__ __forceinline__ void set(const double* __restrict__ p, int64_t off) { b = p + off; }
    __device__ __forceinline__ Raw loadE(int lane, int c) const { return b[2 * lane + 128 * c]; }
    __device__ __forceinline__ Raw loadO(int lane, int c) const { return b[2 * (63 - lane) + 1 + 128 * c]; }
    __device__ __forceinline__ double getE(Raw w, const SampleConv& k) const { return w * k.sgn; }
    __device__ __forceinline__ double getO(Raw w, const SampleConv& k) const { return w * k.sgn; }
};

// pre[lane + 64 r] = pre[lane] * W32^r and post[lane + 64 q] = post[lane] * W32^q (both tables are unit-circle
// points whose angle is affine in the index with step 2 pi/2048 resp. 4 * 2 pi/4096 per index)
template <int R> __device__ __forceinline__ double2 twiddle_lane(double2 v, double2 laneConst) {
    double2 t = cmul(v, laneConst);
    if (R == 0) return t;
    return mul_w32<R == 0 ? 1 : R>(t);
}

// Which samples a lane holds.  The fold of the N windowed samples y into the N/4 complex points of the transform
// (mdct.py:63-76 in its N/4 form) takes, for point n = lane + 64 R,
//     R < 4:   re = -y[2 (767 - n) + 1] - y[2 (768 + n)],   im =  y[2 (255 - n) + 1] - y[2 (256 + n)]
//     R >= 4:  re =  y[2 (n - 256)] - y[2 (767 - n) + 1],   im = -y[2 (256 + n)] - y[2 (1279 - n) + 1]
// i.e. EVEN samples at pair indices that run WITH the lane and ODD samples at pair indices that run AGAINST it.  So a lane
// loads its even samples at pairs lane + 64 c and its odd samples at pairs (63 - lane) + 64 c, c = 0..15 (each wave load
// still covers one contiguous 256-byte / 1-KiB stretch, lanes in reverse order for the odd ones), and the fold is
// arithmetic on the lane's own registers: 767 - n = (63 - lane) + 64 (11 - R), 255 - n = (63 - lane) + 64 (3 - R),
// 1279 - n = (63 - lane) + 64 (19 - R).  No LDS staging of the windowed block, no gather, and a block at an odd sample
// offset costs nothing extra.  The first hop of a block is c = 0..7, the second c = 8..15, for both kinds.
template <int NSIG, bool REUSE, class T>
__global__ __launch_bounds__(kWave * kWavesPerBlock, MRC_MDCT_WG_PER_CU) void mdct_long_kernel(
    DevShape S, int64_t nUnits, const T* __restrict__ chL, const T* __restrict__ chR, int64_t stride,
    const int64_t* __restrict__ offsets, double* __restrict__ lines, int* __restrict__ oscale) {
    __shared__ __attribute__((aligned(16))) double smem[kWavesPerBlock * kWaveLds + 2 * kQ];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // a scalar: unit, signal, offset and every branch on them are wave-uniform
    double* ws = smem + wave * kWaveLds;                               // this wave's exchange / output staging
    double2* w512 = reinterpret_cast<double2*>(smem + kWavesPerBlock * kWaveLds);   // [512] e^{-2 pi i t/512}
    for (int i = threadIdx.x; i < kQ; i += kWave * kWavesPerBlock) w512[i] = S.wQ[i];
    // lane-constant registers: window at the lane's even (forward) samples -- its odd (mirrored) samples 2 (63 - lane) + 1
    // + 128 c are the mirror images N - 1 - n of the even ones of stretch 15 - c, and the long block's window is symmetric
    // bit for bit (DevShape::winSymmetric, a condition of this kernel) --; pre/post twiddle of the lane's first point
    const int evn = 2 * lane;                                          // sample index inside a 128-sample stretch
    double wE[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) wE[c] = S.win[evn + 128 * c];
    const double2 preLane = S.pre[lane];
    const double2 postLane = S.post[lane];
    __syncthreads();                            // W512 table visible to all waves

    // REUSE (hop-overlapped stream): a wave walks CONSECUTIVE FRAMES OF ONE SIGNAL and keeps the raw second hop of its block in
    // registers -- it is the first hop of the next block.  Mono: kRun consecutive units.  Joint: the workgroup takes kRun
    // frames; wave w walks signal w (L, R, M, S) over the first half of them and signal (w + 2) % 4 over the second half,
    // so every wave transforms as many M / S units (which convert BOTH channels) as L / R units whichever SIMD it sits
    // on, and the four waves read the same L / R hops at the same time (one trip from HBM, three L2 hits).
    static_assert(!REUSE || NSIG == 1 || NSIG == kWavesPerBlock, "joint reuse: one wave per signal");
    constexpr bool kJointRuns = REUSE && NSIG != 1;
    constexpr int kSub = kJointRuns ? kRun / 2 : kRun;                 // units a wave walks in one go
    int64_t firstUnit, step;
    if (REUSE && NSIG == 1) { firstUnit = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kRun; step = 1; }
    else if (REUSE) { firstUnit = (int64_t)blockIdx.x * kRun * NSIG + wave; step = NSIG; }
    else { firstUnit = (int64_t)blockIdx.x * kWavesPerBlock * kRun + wave; step = kWavesPerBlock; }

    double rawE[8], rawO[8];                    // REUSE: raw second hop of the previous block (= first hop of this one)
    // REUSE: the new hop of the coming unit, requested one unit ahead -- for int16 PCM (16 registers per channel); float64
    // samples would take 32 per channel, more than the wave has left: they are loaded when their unit starts
#ifndef MRC_MDCT_AHEAD                           // 0: never, 1: joint blocks only, 2: mono and joint
#define MRC_MDCT_AHEAD 2
#endif
    constexpr bool kAhead = sizeof(T) == 2 && (MRC_MDCT_AHEAD == 2 || (MRC_MDCT_AHEAD == 1 && NSIG != 1));
    constexpr int kNxt = kAhead ? 8 : 1;
    typename ChanView<T>::Raw nxtAE[kNxt], nxtAO[kNxt], nxtBE[kNxt], nxtBO[kNxt];        // A: the signal's (first) channel, B: the right channel of M / S
    int64_t prevOff = 0;
    // the it-th unit of this wave (joint runs: the second half of the frames with the other kind of signal)
    auto unit_of = [&](int it) -> int64_t {
        int64_t unit = firstUnit + (int64_t)it * step;
        if (kJointRuns && it >= kSub) unit += (wave < 2 ? 2 : -2);
        return unit;
    };
    // Explicit offsets of the wave's kRun units, fetched ONCE into lane `it` and read back with v_readlane: a vector load
    // inside the loop drains the memory counter it shares with the previous unit's line stores (the store latency of every
    // unit exposed), and so does any wait the compiler derives from one.
    int64_t offLane = 0;
    if (offsets) {
        const int64_t u = unit_of(lane < kRun ? lane : 0);
        if (lane < kRun && u < nUnits) offLane = offsets[NSIG == 1 ? u : u / NSIG];
    }
    auto off_of = [&](int it, int64_t unit) -> int64_t {
        if (!offsets) return (NSIG == 1 ? unit : unit / NSIG) * stride;
        return (int64_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(offLane >> 32), it) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)offLane, it));
    };

    // Every load issued so far (window, twiddles, offsets) lands HERE, on all paths: left pending, the wait for them is
    // emitted at their first use inside the loop, where it is a wait for "everything but the newest N operations" on the
    // counter that loads share with stores in order -- i.e. for the previous unit's line stores, in every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), expcnt / lgkmcnt untouched
    for (int it = 0; it < kRun; ++it) {
        const int64_t unit = unit_of(it);
        if (unit >= nUnits) break;                                     // wave-uniform (a frame has all its signals or none)
        const int64_t off = off_of(it, unit);
        const int sig = NSIG == 1 ? 0 : (int)(unit % NSIG);
        // REUSE: this block continues the previous one of the wave (its first hop is in rawE / rawO and its second hop
        // was requested a unit ago) -- always in a strided stream, and for explicit offsets whenever they are a hop apart
        // (a block-switched stream is mostly runs of long blocks); else the whole block is loaded here
        const bool cont = REUSE && (it % kSub) != 0 && off == prevOff + kM;     // wave-uniform

        // ---- A. the block's samples: first hop kept or loaded, second hop from the request of a unit ago or loaded.
        // L / R units read one channel (chosen by a scalar pointer), M / S units both: the two forms are two copies of this
        // phase behind ONE wave-uniform branch per unit
        const SampleConv kA = sample_conv(false), kB = sample_conv(NSIG != 1 && sig == 3);
        ChanView<T> vA, vB;
        vA.set((NSIG != 1 && sig == 1) ? chR : chL, off);
        if (NSIG != 1) vB.set(chR, off);
        double curE[8], curO[8];
        auto gather = [&](auto bothTag) {
            constexpr bool BOTH = decltype(bothTag)::value;
            auto now = [&](int c, double* e, double* o) {          // stretch c of the block, loaded and converted here
                const auto ae = vA.loadE(lane, c), ao = vA.loadO(lane, c);
                if (BOTH) {
                    const auto be = vB.loadE(lane, c), bo = vB.loadO(lane, c);
                    *e = (vA.getE(ae, kA) + vB.getE(be, kB)) / 2.0;
                    *o = (vA.getO(ao, kA) + vB.getO(bo, kB)) / 2.0;
                } else { *e = vA.getE(ae, kA); *o = vA.getO(ao, kA); }
            };
            if (cont) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (kAhead) {
                        constexpr int kOne = kAhead ? 1 : 0;
                        curE[c] = BOTH ? (vA.getE(nxtAE[c * kOne], kA) + vB.getE(nxtBE[c * kOne], kB)) / 2.0 : vA.getE(nxtAE[c * kOne], kA);
                        curO[c] = BOTH ? (vA.getO(nxtAO[c * kOne], kA) + vB.getO(nxtBO[c * kOne], kB)) / 2.0 : vA.getO(nxtAO[c * kOne], kA);
                    } else now(c + 8, &curE[c], &curO[c]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) now(c, &rawE[c], &rawO[c]);
#pragma unroll
                for (int c = 0; c < 8; ++c) now(c + 8, &curE[c], &curO[c]);
            }
            if (REUSE && kAhead) {   // the next unit's new hop (if it continues this one; if not, it is loaded when its turn comes):
                                     // stretches 16..23 counted from THIS block's start, same word parity
                const int64_t nu = unit + step;
                if ((it + 1) % kSub != 0 && nu < nUnits && off_of(it + 1, nu) == off + kM) {
#pragma unroll
                    for (int c = 0; c < (kAhead ? 8 : 0); ++c) {
                        nxtAE[c] = vA.loadE(lane, c + 16); nxtAO[c] = vA.loadO(lane, c + 16);
                        if (BOTH) { nxtBE[c] = vB.loadE(lane, c + 16); nxtBO[c] = vB.loadO(lane, c + 16); }
                    }
                }
            }
        };
        if (NSIG != 1 && sig >= 2) gather(std::true_type{});
        else gather(std::false_type{});
        prevOff = off;
        // ---- B. window, fold N -> N/2 -> 512 complex points (n = lane + 64 r) in registers, pre-twiddle
        double2 u[8];
#define MRC_YE(C) ((((C) & 15) < 8 ? rawE[(C) & 7] : curE[(C) & 7]) * wE[(C) & 15])     /* (& 15: the branch not taken of MRC_FOLD is still parsed) */
#define MRC_YO(C) ((((C) & 15) < 8 ? rawO[(C) & 7] : curO[(C) & 7]) * wE[15 - ((C) & 15)])
#define MRC_FOLD(R)                                                                     \
        {                                                                               \
            double re, im;                                                              \
            if (R < 4) { re = -MRC_YO(11 - R) - MRC_YE(12 + R); im = MRC_YO(3 - R) - MRC_YE(4 + R); }   \
            else { re = MRC_YE(R - 4) - MRC_YO(11 - R); im = -MRC_YE(4 + R) - MRC_YO(19 - R); }         \
            u[R] = twiddle_lane<R>(make_double2(re, im), preLane);                      \
        }
        MRC_FOLD(0) MRC_FOLD(1) MRC_FOLD(2) MRC_FOLD(3) MRC_FOLD(4) MRC_FOLD(5) MRC_FOLD(6) MRC_FOLD(7)
#undef MRC_FOLD
#undef MRC_YE
#undef MRC_YO
        if (REUSE) {
#pragma unroll
            for (int c = 0; c < 8; ++c) { rawE[c] = curE[c]; rawO[c] = curO[c]; }
        }
        // ---- C. FFT-512 = 8 x 8 x 8
        dft8(u);
        double2* ex = reinterpret_cast<double2*>(ws);
        if (!(MRC_MDCT_PROFILE & 4)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ex[9 * lane + q] = u[q];           // element 8*lane+q, padded (+1 per 8)
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = ex[lane + (lane >> 3) + 72 * r];   // element lane + 64 r
        wave_sync();
        }
#pragma unroll
        for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], w512[8 * (lane & 7) * r]);     // W512^(8 k r), k = lane % 8
        dft8(u);
        {
            const int base = 64 * (lane >> 3) + (lane & 7);
#pragma unroll
            for (int q = 0; q < 8; ++q) if (!(MRC_MDCT_PROFILE & 4)) ex[base + 8 * q] = u[q];
        }
        if (!(MRC_MDCT_PROFILE & 4)) {
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = ex[lane + 64 * r];
        wave_sync();
        }
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
            if (!(MRC_MDCT_PROFILE & 1) || xE[i] == 1.2345e300) *reinterpret_cast<double2*>(dst + 2 * i) = make_double2(xE[i], xO[i]);
        }
        peak = wave_max(peak);
        if (lane == 0) oscale[unit] = scale_factor20(peak, S.nScaleBits);    // codecThem.py:321-322
        wave_sync();               // output staging read before the next unit overwrites the region
    }
}

}  // namespace

bool mdct_long_applicable(const DevShape& S, int64_t stride, const int64_t* offsets, const void* chL, const void* chR,
                          int fmt) {
    if (S.a != 1024 || S.b != 1024 || !S.winSymmetric) return false;
    // samples are loaded one by one (2 or 8 bytes): any stride, any offset; the base needs the sample's own alignment
    const uintptr_t mask = fmt == kSampleI16 ? 1 : 7;
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
