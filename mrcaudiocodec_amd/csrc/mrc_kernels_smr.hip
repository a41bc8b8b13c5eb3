// smr_kernel -- psychoac.py:134-219 on gfx950: Hann window -> real FFT -> intensity spectrum -> tonal
// maskers (strict 3-point peaks, kept in bin order) -> masked threshold on the MDCT line grid ->
// SMR per scale-factor band (and the per-band max |X| the scale factors need).  One 256-thread workgroup per
// (frame, signal), any block shape; units are walked in an XCD-contiguous order.
//
// The cost is the spreading (psychoac.py:68-78,166-168): ~P maskers x N/2 lines of 10^x in float64
// (P ~ 257, N/2 = 1024 on white noise).  Two evaluation modes:
//
//  EXACT = true   the reference's expression, operation by operation, pow() per (masker, line), summed
//                 in masker order.  ~200 fp64 instructions per pair.
//  EXACT = false  (default) "sorted sweep".  Lines and maskers are both sorted in Bark, so for a line k
//                 the maskers split into three index ranges (found once per frame from the masker side):
//                   more than 1/2 Bark below the line: I_m * 2^(s_m (z_k - z_m - 1/2)), s_m the level-dependent
//                           upper slope.  FAR FIELD (maskers below every line of a 64-line chunk): expansion in
//                           (slope - middle slope of the frame) x (distance from the chunk centre), one 2^x per
//                           masker and chunk, order 8/12/16 chosen from a rigorous bound -- far_group().  NEAR
//                           FIELD: one table-driven 2^x per pair (exp2_tab64, 16 instructions);
//                   inside +-1/2 Bark: exactly I_m, as a difference of double-double prefix sums;
//                   more than 1/2 Bark above: the lower slope is the same -27 dB/Bark for every masker, so the sum
//                           factors into one table value per line times a suffix sum over maskers (exponents
//                           carried in double-double).
//                 The band maximum of SPL(line) - SPL(threshold) is taken on the RATIO of the two intensities and
//                 converted with one log10 per band.  Same integers as EXACT on every parity corpus, thresholds
//                 within 1e-10 dB (tests/test_gpu_parity.py::test_spread_modes_agree).
// DESIGN.md section 4 has the derivations, the error bounds and the measured instruction counts.
#include "mrc_device.hpp"

#include <algorithm>
#include <type_traits>
#include "mrc_log10.hpp"

namespace mrc {
using namespace dev;
namespace {

constexpr int kLinesPerThread = 4;                     // EXACT mode register tile

// Order-preserving map double -> uint64 (a < b  <=>  key(a) < key(b), -0 < +0), so that a maximum over
// doubles can be taken with an integer LDS atomic.
__device__ __forceinline__ unsigned long long order_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double order_value(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
// LDS traffic between lanes of ONE wave (smr_short_kernel): order the wave's own DS operations, keep the compiler from
// moving LDS accesses across this point
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// far-field expansion (see the sweep): highest order, fewest maskers worth it, and for each supported order J the
// largest |x| with |x|^(J+1)/(J+1)! e^|x| below 1e-15 (x = slope spread * half the Bark span of a group of lines)
constexpr int kFarMaxOrder = 20;
#ifndef MRC_FAR_MIN
#define MRC_FAR_MIN 24
#endif
constexpr int kFarMinMaskers = MRC_FAR_MIN;
#ifdef MRC_FAR_STRICT       // round-1 limits: truncated tail < 1e-17 of each term
constexpr double kFarLimit8 = 0.052, kFarLimit12 = 0.27, kFarLimit16 = 0.68, kFarLimit20 = 1.0;
#elif defined(MRC_FAR_TOL13)  // experiment: truncated tail < 1e-13 of each term
constexpr double kFarLimit8 = 0.1466, kFarLimit12 = 0.5436, kFarLimit16 = 1.1529, kFarLimit20 = 1.9056;
#else                       // truncated tail < 1e-15 of each term (2 % faster: more chunks get by with a lower order; the
constexpr double kFarLimit8 = 0.089, kFarLimit12 = 0.397, kFarLimit16 = 0.94, kFarLimit20 = 1.3;   // thresholds move < 1e-14 dB)
#endif
#ifndef MRC_FAR_MAX_ORDER                        // 20 costs 48 accumulator registers: spills around every chunk
#define MRC_FAR_MAX_ORDER 16
#endif
constexpr double kFarLimitMax = MRC_FAR_MAX_ORDER >= 20 ? kFarLimit20 : MRC_FAR_MAX_ORDER >= 16 ? kFarLimit16 : kFarLimit12;
constexpr double kInvFactorial[kFarMaxOrder + 1] = {
    1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
    1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
    1.0 / 1307674368000.0, 1.0 / 20922789888000.0, 1.0 / 355687428096000.0, 1.0 / 6402373705728000.0,
    1.0 / 121645100408832000.0, 1.0 / 2432902008176640000.0};
constexpr double kLog2Of10 = 0x1.a934f0979a371p+1;     // log2(10) = hi + lo
constexpr double kLog2Of10Lo = 0x1.7f2495fb7fa6dp-53;
// b = -2.7*log2(10) bits per Bark below the masker (psychoac.py:74), split hi + lo
constexpr double kLowHi = -0x1.1f03bbffee7edp+3;
constexpr double kLowLo = 0x1.e3c74df63d090p-51;

// 2^f on [-0.5, 0.5]: degree-11 Chebyshev-node fit, max relative error 2e-16 including evaluation.
__device__ __forceinline__ double exp2_poly(double f) {
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
    return fma(p, f, 1.0);
}

// 2^(sT*u/T) for the spreading loop, table driven (T = kExpTab entries per octave, sT = slope in 1/T bit per Bark):
// n = rint(sT*u) splits into k = n / T (exponent), j = n mod T (entry of the 2^(j/T) table in LDS) and a remainder
// g = sT*u - n in [-1/2, 1/2] (exact, by fma) whose 2^(g/T) = exp(g ln2/T) is a Taylor polynomial (T = 64: degree 5,
// remainder < 3.5e-17).  sT*u == 0 gives exactly 1 (a line inside +-1/2 Bark
// gets exactly the masker's intensity).  Requires |sT*u| < 2^31 (here it is < 16000).
constexpr int kExpTab = 64;
constexpr int kExpTabShift = 6;
// an SPL reaches its -30 dB floor at an intensity of 10^-12.6 (psychoac.py:8-12); above this guard it does not
constexpr double kSplFloorGuard = 1e-12;
// T = 256 (the long block's sweep, MRC_EXP_TAB_LONG): a table four times as fine takes one term off the polynomial
// (|x ln2 / 256|^5 / 5! < 4e-17 for the remainder |x| <= 1/2).
#ifndef MRC_EXP_TAB_LONG
#define MRC_EXP_TAB_LONG 256
#endif
template <int T = kExpTab>
__device__ __forceinline__ double exp2_tab64(double sT, double u, const double* __restrict__ tab) {
    const double shifter = 0x1.8p52;
    const double tt = fma(sT, u, shifter);
    const double r = tt - shifter;
    const double g = fma(sT, u, -r);
    const int n = __double2loint(tt);
    if (T == 256) {
        double p = fma(0x1.3b2ab6fba4e77p-39, g, 0x1.c6b08d704a0c0p-29);
        p = fma(p, g, 0x1.ebfbdff82c58fp-19);
        p = fma(p, g, 0x1.62e42fefa39efp-9);
        p = fma(p, g, 1.0);
        return ldexp(p * tab[n & 255], n >> 8);
    }
    double p = fma(0x1.5d87fe78a6731p-40, g, 0x1.3b2ab6fba4e77p-31);
    p = fma(p, g, 0x1.c6b08d704a0c0p-23);
    p = fma(p, g, 0x1.ebfbdff82c58fp-15);
    p = fma(p, g, 0x1.62e42fefa39efp-7);
    p = fma(p, g, 1.0);
    // (the table as two arrays of 32-bit halves -- entry j of either in bank j, conflict-free for any index pattern -- was
    // measured in round 3: 4.60 against 4.54 ms; like the 32-entry table of round 2 it removes conflicts the waves do not wait for)
    return ldexp(p * tab[n & (kExpTab - 1)], n >> kExpTabShift);
}

// 2^(j/64), j = 0..63, correctly rounded
__constant__ double kExp2Tab[kExpTab] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0
};

// 2^(j/256), j = 0..255, correctly rounded (the long block's table)
__constant__ double kExp2Tab256[256] = {
    0x1.0000000000000p+0, 0x1.00b1afa5abcbfp+0, 0x1.0163da9fb3335p+0, 0x1.02168143b0281p+0,
    0x1.02c9a3e778061p+0, 0x1.037d42e11bbccp+0, 0x1.04315e86e7f85p+0, 0x1.04e5f72f654b1p+0,
    0x1.059b0d3158574p+0, 0x1.0650a0e3c1f89p+0, 0x1.0706b29ddf6dep+0, 0x1.07bd42b72a836p+0,
    0x1.0874518759bc8p+0, 0x1.092bdf66607e0p+0, 0x1.09e3ecac6f383p+0, 0x1.0a9c79b1f3919p+0,
    0x1.0b5586cf9890fp+0, 0x1.0c0f145e46c85p+0, 0x1.0cc922b7247f7p+0, 0x1.0d83b23395decp+0,
    0x1.0e3ec32d3d1a2p+0, 0x1.0efa55fdfa9c5p+0, 0x1.0fb66affed31bp+0, 0x1.1073028d7233ep+0,
    0x1.11301d0125b51p+0, 0x1.11edbab5e2ab6p+0, 0x1.12abdc06c31ccp+0, 0x1.136a814f204abp+0,
    0x1.1429aaea92de0p+0, 0x1.14e95934f312ep+0, 0x1.15a98c8a58e51p+0, 0x1.166a45471c3c2p+0,
    0x1.172b83c7d517bp+0, 0x1.17ed48695bbc0p+0, 0x1.18af9388c8deap+0, 0x1.1972658375d2fp+0,
    0x1.1a35beb6fcb75p+0, 0x1.1af99f8138a1cp+0, 0x1.1bbe084045cd4p+0, 0x1.1c82f95281c6bp+0,
    0x1.1d4873168b9aap+0, 0x1.1e0e75eb44027p+0, 0x1.1ed5022fcd91dp+0, 0x1.1f9c18438ce4dp+0,
    0x1.2063b88628cd6p+0, 0x1.212be3578a819p+0, 0x1.21f49917ddc96p+0, 0x1.22bdda27912d1p+0,
    0x1.2387a6e756238p+0, 0x1.2451ffb82140ap+0, 0x1.251ce4fb2a63fp+0, 0x1.25e85711ece75p+0,
    0x1.26b4565e27cddp+0, 0x1.2780e341ddf29p+0, 0x1.284dfe1f56381p+0, 0x1.291ba7591bb70p+0,
    0x1.29e9df51fdee1p+0, 0x1.2ab8a66d10f13p+0, 0x1.2b87fd0dad990p+0, 0x1.2c57e39771b2fp+0,
    0x1.2d285a6e4030bp+0, 0x1.2df961f641589p+0, 0x1.2ecafa93e2f56p+0, 0x1.2f9d24abd886bp+0,
    0x1.306fe0a31b715p+0, 0x1.31432edeeb2fdp+0, 0x1.32170fc4cd831p+0, 0x1.32eb83ba8ea32p+0,
    0x1.33c08b26416ffp+0, 0x1.3496266e3fa2dp+0, 0x1.356c55f929ff1p+0, 0x1.36431a2de883bp+0,
    0x1.371a7373aa9cbp+0, 0x1.37f26231e754ap+0, 0x1.38cae6d05d866p+0, 0x1.39a401b7140efp+0,
    0x1.3a7db34e59ff7p+0, 0x1.3b57fbfec6cf4p+0, 0x1.3c32dc313a8e5p+0, 0x1.3d0e544ede173p+0,
    0x1.3dea64c123422p+0, 0x1.3ec70df1c5175p+0, 0x1.3fa4504ac801cp+0, 0x1.40822c367a024p+0,
    0x1.4160a21f72e2ap+0, 0x1.423fb2709468ap+0, 0x1.431f5d950a897p+0, 0x1.43ffa3f84b9d4p+0,
    0x1.44e086061892dp+0, 0x1.45c2042a7d232p+0, 0x1.46a41ed1d0057p+0, 0x1.4786d668b3237p+0,
    0x1.486a2b5c13cd0p+0, 0x1.494e1e192aed2p+0, 0x1.4a32af0d7d3dep+0, 0x1.4b17dea6db7d7p+0,
    0x1.4bfdad5362a27p+0, 0x1.4ce41b817c114p+0, 0x1.4dcb299fddd0dp+0, 0x1.4eb2d81d8abffp+0,
    0x1.4f9b2769d2ca7p+0, 0x1.508417f4531eep+0, 0x1.516daa2cf6642p+0, 0x1.5257de83f4eefp+0,
    0x1.5342b569d4f82p+0, 0x1.542e2f4f6ad27p+0, 0x1.551a4ca5d920fp+0, 0x1.56070dde910d2p+0,
    0x1.56f4736b527dap+0, 0x1.57e27dbe2c4cfp+0, 0x1.58d12d497c7fdp+0, 0x1.59c0827ff07ccp+0,
    0x1.5ab07dd485429p+0, 0x1.5ba11fba87a03p+0, 0x1.5c9268a5946b7p+0, 0x1.5d84590998b93p+0,
    0x1.5e76f15ad2148p+0, 0x1.5f6a320dceb71p+0, 0x1.605e1b976dc09p+0, 0x1.6152ae6cdf6f4p+0,
    0x1.6247eb03a5585p+0, 0x1.633dd1d1929fdp+0, 0x1.6434634ccc320p+0, 0x1.652b9febc8fb7p+0,
    0x1.6623882552225p+0, 0x1.671c1c70833f6p+0, 0x1.68155d44ca973p+0, 0x1.690f4b19e9538p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6b052fa75173ep+0, 0x1.6c012750bdabfp+0, 0x1.6cfdcddd47645p+0,
    0x1.6dfb23c651a2fp+0, 0x1.6ef9298593ae5p+0, 0x1.6ff7df9519484p+0, 0x1.70f7466f42e87p+0,
    0x1.71f75e8ec5f74p+0, 0x1.72f8286ead08ap+0, 0x1.73f9a48a58174p+0, 0x1.74fbd35d7cbfdp+0,
    0x1.75feb564267c9p+0, 0x1.77024b1ab6e09p+0, 0x1.780694fde5d3fp+0, 0x1.790b938ac1cf6p+0,
    0x1.7a11473eb0187p+0, 0x1.7b17b0976cfdbp+0, 0x1.7c1ed0130c132p+0, 0x1.7d26a62ff86f0p+0,
    0x1.7e2f336cf4e62p+0, 0x1.7f3878491c491p+0, 0x1.80427543e1a12p+0, 0x1.814d2add106d9p+0,
    0x1.82589994cce13p+0, 0x1.8364c1eb941f7p+0, 0x1.8471a4623c7adp+0, 0x1.857f4179f5b21p+0,
    0x1.868d99b4492edp+0, 0x1.879cad931a436p+0, 0x1.88ac7d98a6699p+0, 0x1.89bd0a478580fp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8be05bad61778p+0, 0x1.8cf3216b5448cp+0, 0x1.8e06a5e0866d9p+0,
    0x1.8f1ae99157736p+0, 0x1.902fed0282c8ap+0, 0x1.9145b0b91ffc6p+0, 0x1.925c353aa2fe2p+0,
    0x1.93737b0cdc5e5p+0, 0x1.948b82b5f98e5p+0, 0x1.95a44cbc8520fp+0, 0x1.96bdd9a7670b3p+0,
    0x1.97d829fde4e50p+0, 0x1.98f33e47a22a2p+0, 0x1.9a0f170ca07bap+0, 0x1.9b2bb4d53fe0dp+0,
    0x1.9c49182a3f090p+0, 0x1.9d674194bb8d5p+0, 0x1.9e86319e32323p+0, 0x1.9fa5e8d07f29ep+0,
    0x1.a0c667b5de565p+0, 0x1.a1e7aed8eb8bbp+0, 0x1.a309bec4a2d33p+0, 0x1.a42c980460ad8p+0,
    0x1.a5503b23e255dp+0, 0x1.a674a8af46052p+0, 0x1.a799e1330b358p+0, 0x1.a8bfe53c12e59p+0,
    0x1.a9e6b5579fdbfp+0, 0x1.ab0e521356ebap+0, 0x1.ac36bbfd3f37ap+0, 0x1.ad5ff3a3c2774p+0,
    0x1.ae89f995ad3adp+0, 0x1.afb4ce622f2ffp+0, 0x1.b0e07298db666p+0, 0x1.b20ce6c9a8952p+0,
    0x1.b33a2b84f15fbp+0, 0x1.b468415b749b1p+0, 0x1.b59728de5593ap+0, 0x1.b6c6e29f1c52ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.b928cf22749e4p+0, 0x1.ba5b030a1064ap+0, 0x1.bb8e0b79a6f1fp+0,
    0x1.bcc1e904bc1d2p+0, 0x1.bdf69c3f3a207p+0, 0x1.bf2c25bd71e09p+0, 0x1.c06286141b33dp+0,
    0x1.c199bdd85529cp+0, 0x1.c2d1cd9fa652cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c544778fafb22p+0,
    0x1.c67f12e57d14bp+0, 0x1.c7ba88988c933p+0, 0x1.c8f6d9406e7b5p+0, 0x1.ca3405751c4dbp+0,
    0x1.cb720dcef9069p+0, 0x1.ccb0f2e6d1675p+0, 0x1.cdf0b555dc3fap+0, 0x1.cf3155b5bab74p+0,
    0x1.d072d4a07897cp+0, 0x1.d1b532b08c968p+0, 0x1.d2f87080d89f2p+0, 0x1.d43c8eacaa1d6p+0,
    0x1.d5818dcfba487p+0, 0x1.d6c76e862e6d3p+0, 0x1.d80e316c98398p+0, 0x1.d955d71ff6075p+0,
    0x1.da9e603db3285p+0, 0x1.dbe7cd63a8315p+0, 0x1.dd321f301b460p+0, 0x1.de7d5641c0658p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e11676b197d17p+0, 0x1.e264614f5a129p+0, 0x1.e3b333b16ee12p+0,
    0x1.e502ee78b3ff6p+0, 0x1.e653924676d76p+0, 0x1.e7a51fbc74c83p+0, 0x1.e8f7977cdb740p+0,
    0x1.ea4afa2a490dap+0, 0x1.eb9f4867cca6ep+0, 0x1.ecf482d8e67f1p+0, 0x1.ee4aaa2188510p+0,
    0x1.efa1bee615a27p+0, 0x1.f0f9c1cb6412ap+0, 0x1.f252b376bba97p+0, 0x1.f3ac948dd7274p+0,
    0x1.f50765b6e4540p+0, 0x1.f6632798844f8p+0, 0x1.f7bfdad9cbe14p+0, 0x1.f91d802243c89p+0,
    0x1.fa7c1819e90d8p+0, 0x1.fbdba3692d514p+0, 0x1.fd3c22b8f71f1p+0, 0x1.fe9d96b2a23d9p+0
};

// 2^(hi + lo), |lo| << 1
__device__ __forceinline__ double exp2_dd(double hi, double lo) {
    const double k = rint(hi);
    return ldexp(exp2_poly((hi - k) + lo), (int)k);
}

// (hi, lo) += (xh, xl) in double-double (Knuth two-sum on the high parts; |lo| << |hi|)
__device__ __forceinline__ void dd_add(double* hi, double* lo, double xh, double xl) {
    const double s = *hi + xh;
    const double v = s - *hi;
    double e = (*hi - (s - v)) + (xh - v);
    e += *lo + xl;
    const double h = s + e;
    *lo = e - (h - s);
    *hi = h;
}

// Sum each of N per-lane values over the 64 lanes of the wave and leave all N totals in every lane.
// Blocks of W = 16 or 8 values go through a "transposing" butterfly: at every halving step a lane hands HALF of its
// remaining values to its partner and adds the partner's half of the others (W/2 + W/4 + ... + 1 exchanges instead
// of W x 6); plain butterfly steps finish the one value left per lane and W broadcasts distribute the totals.
// Values beyond the last full block are reduced one by one.
// Cross-lane primitives of the coefficient reduction, all register-to-register (no LDS round trips):
//   distance 32 / 16: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper half (odd 16-lane rows) of one
//                     register with the lower half (even rows) of another -- exactly one step of a transposing
//                     butterfly: afterwards the lower lanes hold both halves' `a`, the upper lanes both halves' `b`;
//   distance 8, 4, 2, 1: DPP row rotate / half mirror / quad permutes.
template <int DIST>
__device__ __forceinline__ double wave_xchg_add(double a, double b) {
    static_assert(DIST == 32 || DIST == 16, "swap distance");
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    const uint2v lo = DIST == 32 ? __builtin_amdgcn_permlane32_swap(alo, blo, false, false)
                                 : __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const uint2v hi = DIST == 32 ? __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false)
                                 : __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
// Sum each of 8 per-lane values over the 64 lanes of the wave and leave the 8 totals in every lane (wave-uniform).
// Transposing butterfly: at distances 32, 16, 8 a lane hands HALF of its remaining values to its partner and adds
// the partner's half of the others (4 + 2 + 1 exchanges instead of 8 x 6); three plain steps finish the one value
// left per lane; the lane group [8j, 8j+8) then holds the total of value j.
template <int W>
__device__ __forceinline__ void wave_sum_block(double* v, int lane) {
    static_assert(W == 8, "block of 8 values");
    const double a0 = wave_xchg_add<32>(v[0], v[4]);           // lanes 0..31 keep values 0..3, lanes 32..63 values 4..7
    const double a1 = wave_xchg_add<32>(v[1], v[5]);
    const double a2 = wave_xchg_add<32>(v[2], v[6]);
    const double a3 = wave_xchg_add<32>(v[3], v[7]);
    const double b0 = wave_xchg_add<16>(a0, a2);               // even rows keep the lower pair, odd rows the upper
    const double b1 = wave_xchg_add<16>(a1, a3);
    const double s0 = b0 + dpp_move<0x128>(b0);                // row_ror:8 = lane ^ 8
    const double s1 = b1 + dpp_move<0x128>(b1);
    double s = (lane & 8) ? s1 : s0;
    s += dpp_move<0xB1>(s);                                    // quad_perm [1,0,3,2]
    s += dpp_move<0x4E>(s);                                    // quad_perm [2,3,0,1]
    s += dpp_move<0x141>(s);                                   // row_half_mirror: the other quad of the 8-lane group
#pragma unroll
    for (int j = 0; j < W; ++j)
        v[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), j * (kWave / W)),
                                __builtin_amdgcn_readlane(__double2loint(s), j * (kWave / W)));
}

// blocks of 8 only: a block of 16 would save three exchanges per 16 values but keeps 24 doubles live at once
template <int N>
__device__ __forceinline__ void wave_sum_all(double* v, int lane) {
    constexpr int n8 = N / 8;
#pragma unroll
    for (int i = 0; i < n8; ++i) wave_sum_block<8>(v + 8 * i, lane);
#pragma unroll
    for (int j = 8 * n8; j < N; ++j) {
        v[j] = wave_sum(v[j]);
    }
}

// Far field of one group of lines (see the sweep): maskers [0, nFar) lie more than 1/2 Bark below every line of
// the group.  With c the group's centre, d = z - c, a_m = s_m ln2 the masker's slope and A any reference slope,
//   sum_m I_m 2^(s_m (z - z_m - 1/2)) = exp(A d) sum_m e_m exp((a_m - A) d) = exp(A d) sum_j d^j/j! B_j,
//   e_m = I_m 2^(s_m (c - z_m - 1/2)),  B_j = sum_m e_m (a_m - A)^j.
// Lanes take maskers (ONE 2^x per masker and group instead of one per masker and line), the J+1 coefficients are
// wave-reduced, every line evaluates the polynomial and one 2^x.  The caller picks J from |a_m - A| |d|.
// NB = J + 1 padded to what wave_sum_all reduces cheapest.
template <int J, int NB, int T>
__device__ __forceinline__ double far_group(const double* __restrict__ mt, int nFar, double cq, double slMid,
                                            double d, int lane, const double* __restrict__ e2tab) {
    double B[NB];
    // the first 64 maskers initialise the sums: every lane takes part, a lane past nFar (>= 1) with a zero term
    {
        const int m = min(lane, nFar - 1);
        const double I = mt[4 * m], zm = mt[4 * m + 1], sl = mt[4 * m + 2];
        double term = (lane < nFar) ? I * exp2_tab64<T>(sl, cq - zm, e2tab) : 0.0;   // cq - zm > 0 for m < nFar
        const double da = (sl - slMid) * (0.6931471805599453094 / T);              // slope offset in nats per Bark
#pragma unroll
        for (int j = 0; j <= J; ++j) {
            B[j] = term;
            term *= da;
        }
#pragma unroll
        for (int j = J + 1; j < NB; ++j) B[j] = 0.0;
    }
    for (int m = lane + kWave; m < nFar; m += kWave) {
        const double I = mt[4 * m], zm = mt[4 * m + 1], sl = mt[4 * m + 2];
        double term = I * exp2_tab64<T>(sl, cq - zm, e2tab);
        const double da = (sl - slMid) * (0.6931471805599453094 / T);
#pragma unroll
        for (int j = 0; j <= J; ++j) {
            B[j] += term;
            term *= da;
        }
    }
    // 1/j! goes onto the per-lane partial sums: the wave totals come back as scalars, and a scalar times a constant
    // would need a register copy first
#pragma unroll
    for (int j = 2; j <= J; ++j) B[j] *= kInvFactorial[j];
    wave_sum_all<NB>(B, lane);
    double p = B[J];
#pragma unroll
    for (int j = J - 1; j >= 0; --j) p = fma(p, d, B[j]);
    return p * exp2_tab64<T>(slMid, d, e2tab);
}


// 1/x for a finite positive normal x: hardware estimate + two Newton steps (relative error ~2^-52; NOT the correctly
// rounded quotient -- used by the fast spreading mode only, where one more rounding per masker / line is inside what the
// FFT in front of it already differs from the reference's by; the EXACT mode divides like the reference)
__device__ __forceinline__ double recip_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
// atan(x) for x >= 0 (psychoac.py:27-29's two calls per masker), <= 2 ulp: x <= 1: x Q(x^2), Q of degree 21 from a
// Chebyshev fit in 60-digit arithmetic (tools/make_atan_poly.py); x > 1: pi/2 - atan(1/x).  ~40 instructions against the
// ~90 of the library's.
// atan(t) = t * Q(t^2), 0 <= t <= 1; Q of degree 21 (tools/make_atan_poly.py)
__device__ constexpr double kAtanQ[22] = {0x1.0000000000000p+0, -0x1.5555555555546p-2, 0x1.999999999861ep-3, -0x1.2492492443a94p-3, 0x1.c71c71b1fed92p-4, -0x1.745d1586bfed2p-4, 0x1.3b1398601e89dp-4, -0x1.1110151cb4f09p-4, 0x1.e1d315290f292p-5, -0x1.aed3667a4693ap-5, 0x1.849ab97c0d9e6p-5, -0x1.5eda2e1403e06p-5, 0x1.385c01bcb507ap-5, -0x1.0b657ae92d3e9p-5, 0x1.a91e0c9b2881ep-6, -0x1.2d3ffbb3d4964p-6, 0x1.6c7238a2d8193p-7, -0x1.6773524f49226p-8, 0x1.12060552e1b82p-9, -0x1.2c4eeb1fa7a5bp-11, 0x1.a2865ec94274cp-14, -0x1.156d8b1441eeep-17};
__device__ __forceinline__ double atan_pos(double x) {
    const bool big = x > 1.0;
    const double t = big ? recip_nr(x) : x;
    const double u = t * t;
    double q = kAtanQ[21];
#pragma unroll
    for (int i = 20; i >= 0; --i) q = fma(q, u, kAtanQ[i]);
    const double a = t * q;
    return big ? (0x1.921fb54442d18p+0 - a) + 0x1.1a62633145c07p-54 : a;
}

// kLog10Tab as [j][4] for the LDS copy
struct LogTabDev { double v[kLogTabEntries * 4]; };
constexpr LogTabDev make_log_tab() {
    LogTabDev t{};
    for (int j = 0; j < kLogTabEntries; ++j)
        for (int c = 0; c < 3; ++c) t.v[4 * j + c] = kLog10Tab[j][c];
    return t;
}
__constant__ LogTabDev kLogTabDev = make_log_tab();

// psychoac.py:8-12 with the table-driven log10 (mrc_log10.hpp).  Anything below the smallest normal number -- zero,
// denormals, negative values -- is more than 3000 dB under the -30 dB floor (a NaN ends there too, as with fmax in
// spl_db); +inf stays +inf.
__device__ __forceinline__ double spl_db_tab(double intensity, const double* __restrict__ tab) {
    if (!(intensity >= 0x1p-1022)) return -30.0;
    if (intensity > 0x1.fffffffffffffp+1023) return intensity;
    return fmax(96 + 10 * log10_tab32(intensity, tab), -30.0);
}

// The reference's own per-line formula (psychoac.py:173,212), for lines whose SPL sits on the -30 dB floor and for
// callers that want the thresholds.  Rare on the full path and deliberately OUT OF LINE: inlined, its constants would
// be hoisted out of the sweep loop and cost registers (and scratch traffic) in every frame.
__device__ __attribute__((noinline)) double excess_plain(double t, double a2, int scale, const double* tab, double* thrOut) {
    const double thr = spl_db_tab(t, tab);
    *thrOut = thr;
    return (spl_db_tab(a2, tab) - 6. * scale) - thr;
}

// Where the staged tables sit in the dynamic LDS (offsets in doubles, chosen by launch_smr): the Bark grid of the
// lines for the masker-side searches, the log10 table, the first quadrant of the FFT twiddles (-1: use global).
struct SmrLds { int zbOff, logOff, twOff; };
// The layout for a block of H FFT points, M lines and `last` searched bins; *total = doubles of dynamic LDS.  One function
// for the launcher (any shape) and, evaluated at compile time, for the kernels specialised on the block dimensions.
__host__ __device__ constexpr SmrLds smr_layout(int H, int M, int last, int* totalOut) {
    int total = 4 * H + last + 1;
    const int pkShorts = (last / 2 + 5) & ~3;
    const int piOff = 2 * H + (pkShorts * 2 + 2 * (M + 2) * 2) / 8;          // where piHi starts (kernel layout)
    const int piLen = 2 * (last / 2 + 2);
    const int logLen = kLogTabEntries * 4;
    SmrLds lay{0, 0, -1};
    if (piOff + (piLen > M ? piLen : M) + logLen <= 4 * H) {
        lay.zbOff = piOff;                               // overwritten by the prefix sums after the searches
        lay.logOff = 4 * H - logLen;
    } else {
        total += total & 1;
        lay.zbOff = total;
        lay.logOff = total + M;
        total += M + logLen;
    }
    if ((H & (H - 1)) == 0 && H >= 16) {                 // first quadrant of the FFT twiddles: in the spectrum area
        if (H / 2 <= last + 1) lay.twOff = 4 * H;
        else { total += total & 1; lay.twOff = total; total += H / 2; }
    }
    if (totalOut) *totalOut = total;
    return lay;
}

// Slope nodes: the most maskers a block of DIM lines can take through them.  Rows of kNodeCols doubles for every fourth
// masker (+ row 0) lie between the per-line counts and the log10 table (where the peak bins and the Bark grid were); the masker
// table (4 P) and the in-band prefix sums (2 (P + 1)) share the first FFT buffer with the band keys and the 2^x table (96).
__host__ __device__ constexpr int node_max_maskers(int DIM) {
    const SmrLds lay = smr_layout(DIM, DIM, DIM - 100, nullptr);
    const int qStart = 2 * DIM + (2 * (DIM + 2) * 2) / 8;               // behind cnt / nUp ((DIM + 2) uint16 each)
    if (lay.logOff <= qStart || lay.logOff >= 4 * DIM) return 0;        // (the log10 table is not behind the rows in this layout)
    const int rows = (lay.logOff - qStart) / 18;
    const int byRows = (rows - 1) * 4, byTable = (2 * DIM - 96 - 2) / 6, bySeg = 3 * 26 * 4 - 4;
    int m = byRows < byTable ? byRows : byTable;
    m = m < bySeg ? m : bySeg;
    return m < 0 ? 0 : m;
}
static_assert(node_max_maskers(1024) == 308, "long block: 78 rows");

// inclusive prefix sum over the 64 lanes, in registers: Kogge-Stone inside each 16-lane row with DPP row shifts (lanes
// that would read across the row's start get 0), then the row totals are passed on with row_bcast:15 / row_bcast:31
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_shift_or_zero(int v) {
    // all rows enabled: bound_ctrl supplies the zero of lanes without a source; a partial row mask leaves the other
    // rows' lanes to the prepared zero
    if (ROW_MASK == 0xf) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_or_zero(double v) {
    return __hiloint2double(dpp_shift_or_zero<CTRL, ROW_MASK>(__double2hiint(v)),
                            dpp_shift_or_zero<CTRL, ROW_MASK>(__double2loint(v)));
}
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/) {
    v += dpp_shift_or_zero<0x111, 0xf>(v);              // row_shr:1
    v += dpp_shift_or_zero<0x112, 0xf>(v);              // row_shr:2
    v += dpp_shift_or_zero<0x114, 0xf>(v);              // row_shr:4
    v += dpp_shift_or_zero<0x118, 0xf>(v);              // row_shr:8
    v += dpp_shift_or_zero<0x142, 0xa>(v);              // row_bcast:15 into rows 1 and 3
    v += dpp_shift_or_zero<0x143, 0xc>(v);              // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double wave_incl_scan(double v) {
    v += dpp_shift_or_zero<0x111, 0xf>(v);              // row_shr:1
    v += dpp_shift_or_zero<0x112, 0xf>(v);              // row_shr:2
    v += dpp_shift_or_zero<0x114, 0xf>(v);              // row_shr:4
    v += dpp_shift_or_zero<0x118, 0xf>(v);              // row_shr:8
    v += dpp_shift_or_zero<0x142, 0xa>(v);              // row_bcast:15 into rows 1 and 3
    v += dpp_shift_or_zero<0x143, 0xc>(v);              // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- Slope nodes: the upper-side sum of a whole frame from R prefix sums over the maskers (round 4).
// U_k = sum_{m < nUp_k} I_m 2^(s_m (zq_k - z_m)), zq_k = z_k - 1/2, is a sum of exponentials in the line's Bark value whose
// rates s_m differ from masker to masker -- which is why the lower side (one rate for all) is a suffix sum and this side was
// not.  Interpolating 2^(s d) in the SLOPE at R equispaced nodes sigma_r = sigma_0 - r h (Lagrange weights lambda_r(s_m)) turns
// it into R sums with one rate each:
//     U_k ~= sum_r 2^(sigma_r zq_k) Q_r[nUp_k],     Q_r[n] = sum_{m < n} lambda_r(s_m) I_m 2^(-sigma_r z_m),
// and because the nodes are equispaced, 2^(sigma_r zq) = E0 g^r with E0 = 2^(sigma_0 zq), g = 2^(-h zq): two 2^x and one
// Horner pass over R prefix sums per LINE (the masker side likewise: 2^(-sigma_0 z_m) and 2^(h z_m)), instead of one 2^x per
// (masker, line) pair near the line and an order-16 expansion per chunk far from it.  The nodes span the frame's own slope
// range [min s, max s] plus kNodeMargin spacings on either side (Lagrange interpolation on equispaced nodes is only well
// behaved away from the ends).  The prefix sums are kept for every fourth masker (a row per quad of lanes of the waves that
// compute the terms: <= 78 rows x 18 columns fit where the peak bins and the Bark grid were); the <= 3 maskers between a
// line's row and its nUp are added as direct pairs.
// Error, per line (DESIGN.md section 4 has the derivation): interpolation <= psi* sum_{m < nUp} I_m |prod_r (theta_m - r)| / R!
// with theta_m = (sigma_0 - s_m) / h and psi* = (h R / |sigma_0|)^R e^-R the maximum over the distance of
// (h d ln2)^R 2^(sigma_0 d) (column R of Q carries the sum); rounding <= K eps E0 sum_{m < nUp} Lambda_m I_m 2^(-sigma_0 z_m),
// Lambda_m = sum_r |lambda_r| (column R + 1).  A chunk one of whose lines has  bound > kNodeTol x (its total masked
// intensity)  is evaluated again by the sorted sweep (upper_cold): lines that live on distant loud maskers (beyond a cliff in
// the spectrum) are where the interpolation is weakest.  Frames whose slope range is too wide for R nodes, with fewer than
// kNodeMinMaskers or more than node_max_maskers(DIM) maskers take the sorted sweep as a whole.
constexpr int kNodeR = 16;
constexpr int kNodeMargin = 1;
constexpr int kNodeCols = kNodeR + 2;
constexpr double kNodeHMax = 0.22;                   // node spacing, bit per Bark: the frame's slope range <= 13 x 0.22 = 2.86
constexpr double kNodeHMin = 1e-3;
constexpr int kNodeMinMaskers = 32;
constexpr int kNodeC = 4;                            // maskers per row of the prefix sums (a quad of lanes)
constexpr int kNodeScanSegs = 7;                     // the scan over the rows: two waves, nine columns each, seven lanes per column
constexpr int kNodeSeg = 11;                         // ... rows per lane (78 rows / 7 lanes)
constexpr double kNodeTol = 1e-13;                   // accepted bound on the error of a line's masked intensity (relative)
constexpr double kNodeRoundEps = 8.0 * 0x1p-53;      // K eps: K = 8 covers the measured rounding (tools/rank_proto2.py: <= 1.1)
constexpr double kExpMinus16 = 1.1253517471925912e-07;
static_assert(kNodeR == 16, "psi* below is written for R = 16");
struct NodeWeights { double c[kNodeR]; };
constexpr NodeWeights make_node_weights() {          // 1 / prod_{j != r} (r - j) = (-1)^(R-1-r) / (r! (R-1-r)!)
    NodeWeights w{};
    for (int r = 0; r < kNodeR; ++r) {
        double f = 1.0;
        for (int j = 2; j <= r; ++j) f *= j;
        for (int j = 2; j <= kNodeR - 1 - r; ++j) f *= j;
        w.c[r] = (((kNodeR - 1 - r) & 1) ? -1.0 : 1.0) / f;
    }
    return w;
}
constexpr NodeWeights kNodeW = make_node_weights();

#ifdef MRC_NODE_STATS                            // diagnostics build: how many units / chunks took which evaluation
__device__ unsigned long long gNodeStats[4];     // units with nodes, units without, chunks by nodes, chunks sent back
#define MRC_NODE_COUNT(i) do { if (lane == 0) atomicAdd(&gNodeStats[i], 1ull); } while (0)
#else
#define MRC_NODE_COUNT(i) do { } while (0)
#endif

#ifndef MRC_PROFILE_SKIP                         // profiling aid (wrong results): bit mask of sweep parts to leave out,
#define MRC_PROFILE_SKIP 0                       // 1 far field, 2 direct pairs, 4 partial pairs, 8 chunk tail
#endif
#ifndef MRC_PROFILE_NODESKIP                     // profiling aid (wrong results): 1 no node terms / row scan, 2 no remainder pairs
#define MRC_PROFILE_NODESKIP 0                   // (any value also switches the error-bound fallback off)
#endif
#ifndef MRC_DIRECT_UNROLL                        // pairs in flight per lane in the direct loops
#define MRC_DIRECT_UNROLL 4
#endif

// Slope nodes, one line: U = E0 Horner_g(row[0 .. R-1]) + the NREM maskers between the line's row and its nUp as direct pairs
// (rem <= NREM of them count).  NREM is a template parameter so that the pairs' loads and the row's are all in flight together
// (a loop with an early exit serialises two dependent LDS round trips per pair).  The Horner pass runs as two chains in g^2.
template <int NREM, int TAB>
__device__ __forceinline__ double node_line(const double* __restrict__ row, const double* __restrict__ mt,
                                            const double* __restrict__ e2tab, int mBase, int rem, int mLast, double zq,
                                            double E0, double g) {
    double I[NREM > 0 ? NREM : 1], zm[NREM > 0 ? NREM : 1], sl[NREM > 0 ? NREM : 1];
#pragma unroll
    for (int j = 0; j < NREM; ++j) {
        const int m = min(mBase + j, mLast);
        I[j] = mt[4 * m]; zm[j] = mt[4 * m + 1]; sl[j] = mt[4 * m + 2];
    }
    const double g2 = g * g;
    double ev = row[kNodeR - 2], od = row[kNodeR - 1];
#pragma unroll
    for (int r = kNodeR - 4; r >= 0; r -= 2) {
        ev = fma(ev, g2, row[r]);
        od = fma(od, g2, row[r + 1]);
    }
    double up = fma(od, g, ev) * E0;
#pragma unroll
    for (int j = 0; j < NREM; ++j) up = fma(j < rem ? I[j] : 0.0, exp2_tab64<TAB>(sl[j], zq - zm[j], e2tab), up);
    return up;
}

// Sorted sweep, far field of chunk c: maskers [0, nFar) lie more than 1/2 Bark below EVERY line of the chunk; their sum is
// evaluated by far_group() for the whole chunk (one group) or its two halves.  The expansion is in (slope - middle slope of
// the frame) x (distance from the group's centre): the order follows from half the slope range times half the Bark span, so a
// frame of similar maskers (noise) gets by with a low order even where 64 lines span more than a Bark, and a frame with a loud
// and a quiet region still qualifies at the top of the spectrum.  false: the chunk takes no far field.
template <int TAB, bool HAVE_FAR>
__device__ __forceinline__ bool far_eval(const double* __restrict__ mt, const double* __restrict__ e2tab,
                                         const double* __restrict__ zbG, int M, int c, int lane, int nFar, double z,
                                         double slMid, double spreadHalf, double* out) {
    if (!HAVE_FAR || nFar < kFarMinMaskers || (MRC_PROFILE_SKIP & 1)) return false;
    // the group geometry is wave-uniform: scalar loads of the chunk's first / middle / last Bark values
    const int kFirst = c * kWave;
    const double zFirst = zbG[kFirst], zLast = zbG[min(kFirst + kWave - 1, M - 1)];
    const double zHalfEnd = zbG[min(kFirst + kWave / 2 - 1, M - 1)], zHalfBeg = zbG[min(kFirst + kWave / 2, M - 1)];
    double need = spreadHalf * (0.5 * (zLast - zFirst));
    int nGroups = 1;
    if (need > kFarLimitMax) {                     // (wave-uniform)
        need = spreadHalf * (0.5 * fmax(zHalfEnd - zFirst, zLast - zHalfBeg));
        nGroups = 2;
    }
    const int order = need <= kFarLimit8 ? 8 : need <= kFarLimit12 ? 12 :
                      (MRC_FAR_MAX_ORDER >= 16 && need <= kFarLimit16) ? 16 :
                      (MRC_FAR_MAX_ORDER >= 20 && need <= kFarLimit20) ? 20 : 0;
    if (!order) return false;
    const int myGroup = (nGroups == 2) ? (lane >> 5) : 0;
    double acc = 0.0;
    for (int g = 0; g < nGroups; ++g) {
        const double cg = (nGroups == 1) ? 0.5 * (zFirst + zLast)
                                         : (g == 0 ? 0.5 * (zFirst + zHalfEnd) : 0.5 * (zHalfBeg + zLast));
        const double cq = cg - 0.5, d = z - cg;
        double p;
        if (order == 8) p = far_group<8, 9, TAB>(mt, nFar, cq, slMid, d, lane, e2tab);
        else if (order == 12) p = far_group<12, 16, TAB>(mt, nFar, cq, slMid, d, lane, e2tab);
#if MRC_FAR_MAX_ORDER >= 20
        else if (order == 20) p = far_group<20, 24, TAB>(mt, nFar, cq, slMid, d, lane, e2tab);
#endif
#if MRC_FAR_MAX_ORDER >= 16
        else if (order == 16) p = far_group<16, 17, TAB>(mt, nFar, cq, slMid, d, lane, e2tab);
#endif
        else p = 0.0;
        if (g == myGroup) acc = p;
    }
    *out = acc;
    return true;
}

// Sorted sweep, near field of a chunk: the maskers [mFirst, max nUp) one 2^x per (masker, line) pair, added to tot.  Lines the
// masker is not below (u = 0) get exactly I_m when they see it at all (m < cnt): the in-band sum of the chunk's tail then
// starts at max nUp.
template <int TAB>
__device__ __forceinline__ double near_eval(const double* __restrict__ mt, const double* __restrict__ e2tab, int nUp,
                                            int cnt, double zq, bool tookFar, double tot) {
    // both counts are non-decreasing in the line index: the chunk's bounds sit in its first and last lane
    const int mLow = __builtin_amdgcn_readfirstlane(cnt);                      // min cnt
    const int mExp = __builtin_amdgcn_readlane(nUp, kWave - 1);                // max nUp
    const int mPlain = min(mExp, mLow);
    const int mFirst = tookFar ? __builtin_amdgcn_readfirstlane(nUp) : 0;
    // some line of the chunk is above the masker's band, every line sees the masker.  Maskers below
    // nUp of the chunk's FIRST line are more than 1/2 Bark below every line: u > 0 without the clamp.
    {
        const int mPos = min(max(__builtin_amdgcn_readfirstlane(nUp), mFirst), mPlain);
        const int mStop = (MRC_PROFILE_SKIP & 2) ? 0 : mPlain;
#pragma unroll MRC_DIRECT_UNROLL
        for (int m = mFirst; m < min(mPos, mStop); ++m) {
            const double I = mt[4 * m], zm = mt[4 * m + 1], sl = mt[4 * m + 2];
            tot = fma(I, exp2_tab64<TAB>(sl, zq - zm, e2tab), tot);
        }
#pragma unroll MRC_DIRECT_UNROLL
        for (int m = mPos; m < mStop; ++m) {
            const double I = mt[4 * m], zm = mt[4 * m + 1], sl = mt[4 * m + 2];
            const double u = fmax(zq - zm, 0.0);
            tot = fma(I, exp2_tab64<TAB>(sl, u, e2tab), tot);
        }
    }
    // same, but part of the chunk lies below the masker's band (only when the chunk spans > 1 Bark)
    for (int m = mPlain; m < ((MRC_PROFILE_SKIP & 4) ? 0 : mExp); ++m) {
        const double I = mt[4 * m], zm = mt[4 * m + 1], sl = mt[4 * m + 2];
        const double u = fmax(zq - zm, 0.0);
        tot = fma(m < cnt ? I : 0.0, exp2_tab64<TAB>(sl, u, e2tab), tot);
    }
    return tot;
}

// The sorted sweep's upper side for ONE chunk, out of line: where the slope-node evaluation sends a chunk back (rare)
template <int TAB>
__device__ __attribute__((noinline)) double upper_cold(const double* mt, const double* e2tab, const double* zbG, int M, int c,
                                                       int lane, int nUp, int cnt, double z, double slMid,
                                                       double spreadHalf) {
    double far = 0.0;
    const bool took = far_eval<TAB, true>(mt, e2tab, zbG, M, c, lane, __builtin_amdgcn_readfirstlane(nUp), z, slMid,
                                          spreadHalf, &far);
    return near_eval<TAB>(mt, e2tab, nUp, cnt, z - 0.5, took, took ? far : 0.0);
}


#ifdef MRC_PROFILE_PHASES
// profiling build only (make EXTRA=-DMRC_PROFILE_PHASES): shader-clock cycles per kernel phase, summed over the waves of
// every 64th workgroup (per workgroup in LDS, flushed once at its end: an atomic to global memory per marker from every wave
// made the build sixteen times slower than the kernel it was meant to describe)
__device__ unsigned long long gPhaseCycles[32];
#define MRC_PHASE(i)                                                                  \
    do {                                                                              \
        const long long now_ = __builtin_readcyclecounter();                          \
        if (lane == 0) atomicAdd(&sPhase_[i], (unsigned long long)(now_ - tPhase_));  \
        tPhase_ = __builtin_readcyclecounter();                                       \
    } while (0)
#else
#define MRC_PHASE(i) do { } while (0)
#endif
#ifdef MRC_PROFILE_STOP                         // profiling aid: leave the kernel after phase MRC_PROFILE_STOP
#define MRC_STOP(i) do { if (MRC_PROFILE_STOP == (i)) return; } while (0)
#else
#define MRC_STOP(i) do { } while (0)
#endif

#ifndef MRC_SMR_WAVES_PER_EU                     // 4 workgroups of 4 waves per CU (what the LDS footprint allows): <= 128 VGPRs
#define MRC_SMR_WAVES_PER_EU 4
#endif
// short blocks (two waves, ~5 KB of LDS per workgroup) are latency-bound: more waves.  Measured per 114 688 short units: 4 waves
// per SIMD 0.714 ms, 5: 0.657, 6: 0.627, 8: 0.612 -- but at 8 (64 registers) 16 registers spill and the scratch traffic is
// 1 GB per step of configs[3] (PMC WRITE_SIZE); 6 (77 registers) spills none.
#ifndef MRC_SMR_WAVES_PER_EU_SHORT
#define MRC_SMR_WAVES_PER_EU_SHORT 6
#endif
#if MRC_SMR_WAVES_PER_EU > 0
#define MRC_SMR_OCC __attribute__((amdgpu_waves_per_eu(DIM == 128 ? MRC_SMR_WAVES_PER_EU_SHORT : MRC_SMR_WAVES_PER_EU, \
                                                       DIM == 128 ? MRC_SMR_WAVES_PER_EU_SHORT : MRC_SMR_WAVES_PER_EU)))
#else
#define MRC_SMR_OCC
#endif

#ifndef MRC_FRONT_PRIO                           // issue priority (0..3) of a wave until it enters the sweep
#define MRC_FRONT_PRIO 3
#endif
#ifndef MRC_FAR_PRIO                             // ... and during the far-field pass (shuffle-heavy reductions)
#define MRC_FAR_PRIO 0
#endif

// DIM: 1024 = the long block (N = 2048: H = M = 1024, 924 bins searched for peaks), 128 = the short block (N = 256: H = M =
// 128, 28 bins), 576 = the transition blocks (N = 1152) with their dimensions as compile-time constants -- loop bounds, index splits and the LDS layout fold into
// immediates; same arithmetic, same results.  0: any shape, dimensions from DevShape.
// MODE: what the hot paths fix at compile time -- 1: mono (one signal per frame, every band wanted, no thresholds out, band
// peaks out); 2: joint stereo with the M/S switch known (four signals, the rest alike); 0: all of it at run time.
template <bool EXACT, class SampleT, int NT, int DIM, int MODE>
__device__ __forceinline__ void smr_body(DevShape S, int nsigArg, const SampleT* __restrict__ chL,
                                         const SampleT* __restrict__ chR, int64_t stride,
                                         const int64_t* __restrict__ offsetsArg, const double* __restrict__ lines,
                                         const int* __restrict__ oscale, double* __restrict__ smr,
                                         double* __restrict__ threshArg, double* __restrict__ bandPeakArg,
                                         const int* __restrict__ msSwitch, SmrLds layArg,
                                         unsigned long long* __restrict__ sens) {
    extern __shared__ double smem[];
    const SmrLds lay = DIM ? smr_layout(DIM, DIM, DIM - 100, nullptr) : layArg;
    __shared__ int waveCnt[NT / kWave];
    // per-band running max of the excess (order-preserving key), per-band max |X| (the bit pattern of |x| orders like |x|).
    __shared__ unsigned long long bandKey[kMaxBands], peakKey[kMaxBands];
    __shared__ unsigned long long slopeKey[2];          // min / max upper slope over the frame's maskers (keys)
    __shared__ unsigned char needBand[kMaxBands];       // joint blocks: does the encoder use THIS signal's SMR of the band?
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    constexpr bool LONG = DIM == 1024;
    const int nsig = MODE == 1 ? 1 : MODE == 2 ? 4 : nsigArg;
    const bool haveSwitch = MODE == 1 ? false : MODE == 2 ? true : msSwitch != nullptr;
    const int64_t* offsets = offsetsArg;                 // (strided frames or explicit block offsets: one select per unit either way)
    double* thresh = MODE ? nullptr : threshArg;
    double* bandPeak = bandPeakArg;
    const bool wantPeak = MODE ? true : bandPeakArg != nullptr;
    const int H = DIM ? DIM : S.H, M = DIM ? DIM : S.halfN;
    const int last = DIM ? DIM - 100 : S.peakLast;      // bins 0 .. last-1 are inspected (psychoac.py:160)
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so hardware block
    // i + 1 runs on another die than block i.  Unit u below is chosen such that every XCD walks a CONTIGUOUS range of
    // (frame, signal) units: neighbouring frames share a hop, and the four signals of a joint frame share all their
    // samples -- with this order the second reader finds them in its own L2 instead of fetching them from HBM again.
    // The four signals of a joint frame cost differently (the sweep skips what the M/S switch does not use), and the
    // hardware deals consecutive workgroups to the shader engines round-robin: with sig = unit % 4 every engine would see
    // ONE signal only and the kernel would wait for the engines with the expensive ones.  Rotating the signals from frame
    // to frame gives every engine the same mix.
    const unsigned slot = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t f = slot / nsig;
#ifdef MRC_SMR_NO_ROTATE
    const int sig = slot % nsig;
#else
    const int sig = (int)((slot + f) % nsig);
#endif
    const unsigned unit = (unsigned)(f * nsig + sig);
    // A joint unit NONE of whose bands the M/S switch selects (all bands M/S: the L and R units; all bands L/R: the M and S
    // units -- the rule for strongly correlated and for unrelated channels) has no reader at all: neither its SMRs nor its band
    // peaks reach the bit allocation or the scale factors (ms_stereo.py:70-81; mrc_kernels_alloc.hip reads the selected
    // signal of every band only).  It ends here, before its first load; its outputs stay unwritten.  Every wave takes the
    // same decision from the same 25 flags: no barrier.
    if (haveSwitch && !thresh) {
        const bool need = lane < S.nBands && ((sig >= 2) == (msSwitch[f * S.nBands + lane] != 0));
        if (!__any(need)) return;
    }
    const int64_t off = offsets ? offsets[f] : f * stride;
    double2* A = (double2*)smem;                        // [H]
    double2* B = A + H;                                 // [H]
    double* xi = smem + 4 * H;    // [peakLast + 1] intensity spectrum; later the suffix sums
    // region B is free once the spectrum is in xi: peak bins, then per-line masker counts (filled below)
    unsigned short* cntArr = reinterpret_cast<unsigned short*>(smem + 2 * H);   // [M + 1]
    unsigned short* nUpArr = cntArr + (M + 2);                           // [M + 1]
    short* pkBin = reinterpret_cast<short*>(nUpArr + (M + 2));           // [<= peakLast/2 + 1] peak bins, increasing; dead after
                                                                         // the masker table (then the start of the node rows)
    double* piHi = reinterpret_cast<double*>(pkBin + ((last / 2 + 5) & ~3));   // [<= peakLast/2 + 2] prefix sums of
    double* piLo = piHi + (last / 2 + 2);                //   the masker intensities, double-double (hi, lo)

#ifdef MRC_PROFILE_PHASES
    __shared__ unsigned long long sPhase_[32];
    if (threadIdx.x < 32) sPhase_[threadIdx.x] = 0ull;
    __syncthreads();
    long long tPhase_ = __builtin_readcyclecounter();
#endif
    // The phases before the sweep are chains of short instruction bursts between barriers and memory waits; the sweep
    // is one long stream of VALU work.  Waves of the four workgroups that share a SIMD are in different phases: the
    // ones in the latency-bound part get issue priority, so their chain is not stretched by a neighbour's sweep.
    __builtin_amdgcn_s_setprio(MRC_FRONT_PRIO);
    if (tid < kMaxBands) bandKey[tid] = 0ull; // below every key; visible after the first barrier
    if (tid < 2) slopeKey[tid] = tid ? 0ull : ~0ull;
    if (tid < kMaxBands) peakKey[tid] = 0ull;
    // ms_stereo.py:70-81 (OverallSMRs) keeps, per band, either the L / R pair of SMRs or the M / S pair: the other two
    // never reach the bit allocation.  With the switch known (it only needs the MDCT lines) the sweep below leaves out
    // the 64-line chunks none of whose bands want this signal -- half of all (signal, band) pairs of a stereo frame.
    if (tid < kMaxBands)
        needBand[tid] = (!haveSwitch || tid >= S.nBands) ? 1 : (((sig >= 2) == (msSwitch[f * S.nBands + tid] != 0)) ? 1 : 0);
    const double* zbS = smem + lay.zbOff;               // staged after the FFT (the area is FFT scratch / dead)
    // 2^(j/T): T = 64 in the tail of region A, behind the masker table; the long block's sweep: T = 256, in the half of the
    // spectrum area the suffix sums leave free (staged when the spectrum is dead, with the scans)
    constexpr int TAB = (DIM == 1024 && !EXACT && NT == 256) ? MRC_EXP_TAB_LONG : kExpTab;
    constexpr int kTabLongOff = 464;                    // (doubles behind the start of the spectrum area; sc takes <= 462)
    const double* e2tab = TAB == kExpTab ? smem + 2 * H - kExpTab : smem + 4 * H + kTabLongOff;
    // per-band max of (line intensity / masked threshold) as the bit pattern of a positive double; in front of e2tab
    unsigned long long* ratioKey = reinterpret_cast<unsigned long long*>(smem + 2 * H - kExpTab - kMaxBands);
    const double* logTabLds = smem + lay.logOff;
    const double* logTab = logTabLds;
    // Hann window (window.py:28-45) and real FFT through an H = N/2 point complex FFT.  All global loads of a
    // thread are issued before the first use: one memory round trip per phase instead of one per iteration.
    constexpr int kPre = 4;
    // (even, odd) sample pairs come as ONE load each when the block starts at an even sample of an aligned channel
    const bool pairAligned = !(off & 1) && !(reinterpret_cast<uintptr_t>(chL) & (2 * sizeof(SampleT) - 1)) &&
                             (!chR || !(reinterpret_cast<uintptr_t>(chR) & (2 * sizeof(SampleT) - 1)));
    // long blocks: a thread's four samples are the inputs of its first butterfly and stay in registers (fft_regs_1024)
    MRC_PHASE(16);
#ifndef MRC_SPLIT_PAIRS
#define MRC_SPLIT_PAIRS 1
#endif
    constexpr bool kFftRegs = LONG && NT == 256 && kPre == 4;
    constexpr bool kSplitPairs = kFftRegs && MRC_SPLIT_PAIRS;
    [[maybe_unused]] double2 fftIn[4];
    [[maybe_unused]] Tw3 fftW1;
    if constexpr (kFftRegs) fftW1 = fft1024_twiddles(S.fftTw, 1, tid);
    for (int n0 = tid; n0 < H; n0 += NT * kPre) {
        double e[kPre], o[kPre], he[kPre], ho[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int n = min(n0 + u * NT, H - 1);
            const double2 eo = load_signal_pair(chL, chR, off + 2 * n, sig, pairAligned);
            e[u] = eo.x;
            o[u] = eo.y;
            he[u] = S.hann[2 * n];
            ho[u] = S.hann[2 * n + 1];
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int n = n0 + u * NT;
            if constexpr (kFftRegs) fftIn[u] = make_double2(e[u] * he[u], o[u] * ho[u]);
            else if (n < H) A[n] = make_double2(e[u] * he[u], o[u] * ho[u]);
        }
    }
    const double xiInv = 1.0 / S.xiDen;
    // Constants that are only needed after the FFT are requested BEFORE it (their LDS homes are FFT scratch until
    // then): the loads complete under the FFT's barriers instead of adding a memory round trip of their own.
    double2 wnPre[kPre];
    double zbPre[kPre], logPre = 0.0, e2Pre = 0.0;
    [[maybe_unused]] double e2Pre64 = 0.0;              // long blocks: the 64-entry table too (the node terms are built while
                                                        // the 256-entry one is being staged)
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        // (long blocks: the real split below works on the pairs (k, H - k), k = tid + 1, tid + 1 + NT)
        if (!kSplitPairs || u < 2) wnPre[u] = S.wN[kSplitPairs ? tid + 1 + u * NT : min(tid + u * NT, last - 1)];
        zbPre[u] = EXACT ? 0.0 : S.zb[min(tid + u * NT, M - 1)];
    }
    if (!EXACT) {
        logPre = kLogTabDev.v[tid & (kLogTabEntries * 4 - 1)];
        e2Pre = TAB == kExpTab ? kExp2Tab[tid & (kExpTab - 1)] : kExp2Tab256[tid & 255];
        if (TAB != kExpTab) e2Pre64 = kExp2Tab[tid & (kExpTab - 1)];
    }
    double2* T;
    if constexpr (kFftRegs) {
        MRC_PHASE(0); MRC_STOP(0);
        T = fft_regs_1024(fftIn, A, B, S.fftTw, fftW1, tid);
    } else if (lay.twOff >= 0) {
        double2* Wq = reinterpret_cast<double2*>(smem + lay.twOff);
        for (int t = tid; t < H / 4; t += NT) Wq[t] = S.wH[t];
        __syncthreads();
        MRC_PHASE(0); MRC_STOP(0);
        if (LONG && NT == 256) T = fft_lds_1024<NT>(A, B, Wq, tid);
        else if (DIM == 128) T = fft_lds_128<NT>(A, B, Wq, tid);
        else
        T = fft_lds_pow2<NT>(A, B, H, S.radH, S.nRadH, TwQuarter{Wq, H / 4 - 1, 31 - __clz(H / 4)}, tid);
    } else {
        __syncthreads();
        MRC_PHASE(0); MRC_STOP(0);
        if (DIM == 576) T = fft_lds_576<NT>(A, B, S.wH, tid);
        else
        T = fft_lds_global<NT>(A, B, H, S.radH, S.nRadH, S.wH, tid);
    }
    MRC_PHASE(1); MRC_STOP(1);
    if constexpr (kSplitPairs) {
        // Bins k and H - k come from the same two values of T: with Xe = (T[k] + conj T[H-k]) / 2, Xo = (T[k] - conj T[H-k]) / 2i
        // and P = w_k Xo, X[k] = Xe + P and X[H-k] = conj(Xe - P) (w_{H-k} = -conj w_k).  A thread takes two pairs -- half the
        // reads of T and one complex product for two bins; bin k exactly as below, bin H - k as below with the mirrored twiddle.
        auto intensity = [&](double2 X) {
            return EXACT ? 4. * (X.x * X.x + X.y * X.y) / S.xiDen : (4. * (X.x * X.x + X.y * X.y)) * xiInv;   // psychoac.py:151
        };
        auto split = [&](int k, double2 w, bool both) {
            const double2 zk = T[k];
            double2 zc = T[(H - k) & (H - 1)];
            zc.y = -zc.y;
            const double2 ev = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
            const double2 d = make_double2(zk.x - zc.x, zk.y - zc.y);
            const double2 od = make_double2(0.5 * d.y, -0.5 * d.x);
            const double2 P = cmul(w, od);
            xi[k] = intensity(make_double2(P.x + ev.x, P.y + ev.y));
            if (both) xi[H - k] = intensity(make_double2(ev.x - P.x, ev.y - P.y));
        };
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = tid + 1 + u * NT;              // 1 .. H / 2
            split(k, wnPre[u], H - k < last && k != H / 2);
        }
        if (tid == 0) split(0, make_double2(1.0, 0.0), false);
    } else
    for (int k0 = tid; k0 < last; k0 += NT * kPre) {
        double2 wn[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) wn[u] = (k0 == tid) ? wnPre[u] : S.wN[min(k0 + u * NT, last - 1)];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int k = k0 + u * NT;
            if (k < last) {
                double2 zk = T[k];
                double2 zc = T[(H - k) % H];
                zc.y = -zc.y;
                double2 ev = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
                double2 d = make_double2(zk.x - zc.x, zk.y - zc.y);
                double2 od = make_double2(0.5 * d.y, -0.5 * d.x);
                double2 X = cmul(wn[u], od);
                X.x += ev.x; X.y += ev.y;
                xi[k] = EXACT ? 4. * (X.x * X.x + X.y * X.y) / S.xiDen      // psychoac.py:151
                              : (4. * (X.x * X.x + X.y * X.y)) * xiInv;     // (one rounding more)
            }
        }
    }
    MRC_PHASE(20);
    __syncthreads();                                    // T (in A or B) is dead from here on
    MRC_PHASE(2); MRC_STOP(2);
    if (!EXACT) {                                       // stage the Bark grid and the log10 table (used after 2 barriers)
        {
            double* zw = smem + lay.zbOff;
#pragma unroll
            for (int u = 0; u < kPre; ++u)
                if (tid + u * NT < M) zw[tid + u * NT] = zbPre[u];
            for (int k = tid + kPre * NT; k < M; k += NT) zw[k] = S.zb[k];
        }
        if (tid < kLogTabEntries * 4) smem[lay.logOff + tid] = logPre;
        if (tid < kExpTab) smem[2 * H - kExpTab + tid] = TAB == kExpTab ? e2Pre : e2Pre64;
        if (tid < kMaxBands) ratioKey[tid] = 0ull;
    }

    // tonal maskers: strict 3-point peaks at bins p = 1 .. last-2, kept in increasing bin order.
    // Table (aliases A), 4 doubles per masker:
    //   EXACT: {level-15 dB, Bark z, 0.37*max(level-40,0), -}
    //   fast : {I = 10^((level-15-96)/10), Bark z, upper slope in bits/Bark, I * 2^(b z)}
    double* mt = smem;
    const int nCand = last - 2;
    const int per = (nCand + NT - 1) / NT;
    const int p0 = 1 + tid * per;
    const int p1 = min(p0 + per, last - 1);
    // a thread's candidate bins and their neighbours are read ONCE (per + 2 values); the peak flags serve the count, the
    // ordered compaction behind the barrier and (MRC_OPT_SENSITIVITY) the near-tie count
    constexpr int kPerMax = 4;                           // long block: 4 candidates per thread, transition: 2, short: 1
    int mine = 0;
    unsigned flags = 0;
    if (per <= kPerMax) {
        double v[kPerMax + 2];
#pragma unroll
        for (int j = 0; j < kPerMax + 2; ++j) v[j] = xi[min(p0 - 1 + j, last - 1)];
#pragma unroll
        for (int j = 0; j < kPerMax; ++j)
            if (p0 + j < p1 && v[j + 1] > v[j] && v[j + 1] > v[j + 2]) flags |= 1u << j;
        mine = __popc(flags);
        if (sens) {
            // MRC_OPT_SENSITIVITY: strict comparisons of psychoac.py:162 that a relative change of kPeakGuard in a bin would
            // turn round (a bin within the guard of a neighbour it has to beat, while it does not clearly lose against the other)
            const double kPeakGuard = 1e-11 * __longlong_as_double((long long)sens[7]);    // (sens[7]: guard scale, 1 or 1e8)
            int near = 0;
#pragma unroll
            for (int j = 0; j < kPerMax; ++j) {
                const double c = v[j + 1], l = v[j], r = v[j + 2];
                const bool nl = fabs(c - l) <= kPeakGuard * c, nr = fabs(c - r) <= kPeakGuard * c;
                near += (p0 + j < p1 && ((nl && (c > r || nr)) || (nr && (c > l || nl)))) ? 1 : 0;
            }
            if (near) atomicAdd(&sens[3], (unsigned long long)near);
        }
    } else {
        for (int p = p0; p < p1; ++p) mine += (xi[p] > xi[p - 1] && xi[p] > xi[p + 1]) ? 1 : 0;
    }
    const int incl = wave_incl_scan(mine, lane);
    if (lane == kWave - 1) waveCnt[wave] = incl;
    MRC_PHASE(17);
    __syncthreads();
    MRC_PHASE(18);
    int before = incl - mine, nPeaks = 0;
    for (int w = 0; w < NT / kWave; ++w) {
        const int c = waveCnt[w];
        if (w < wave) before += c;
        nPeaks += c;
    }
    // compact the peak bins first (ordered), then one masker per thread: the transcendental-heavy
    // table entry is computed by full waves instead of the few lanes that happen to own a peak
    if (per <= kPerMax) {
#pragma unroll
        for (int j = 0; j < kPerMax; ++j)
            if ((flags >> j) & 1u) pkBin[before++] = (short)(p0 + j);
    } else {
        for (int p = p0; p < p1; ++p)
            if (xi[p] > xi[p - 1] && xi[p] > xi[p + 1]) pkBin[before++] = (short)p;
    }
    if (!EXACT) {                                        // the two count histograms (adjacent: 2 (M + 2) shorts), eight bytes a store
        unsigned long long* z = reinterpret_cast<unsigned long long*>(cntArr);
        for (int k = tid; k < (M + 2) / 2; k += NT) z[k] = 0ull;
    }
    MRC_PHASE(19);
    __syncthreads();
    MRC_PHASE(3); MRC_STOP(3);
    double slLo = 1e300, slHi = -1e300;                 // this thread's maskers: range of the upper slope
    for (int mi = tid; mi < nPeaks; mi += NT) {
        const int before = mi;
        const int p = pkBin[mi];
        const double x0 = xi[p - 1], x1 = xi[p], x2 = xi[p + 1];
        {
            double s3 = (x0 + x1) + x2;
            MRC_PHASE(21);
            double level = EXACT ? spl_db(s3) : spl_db_tab(s3, logTabLds);   // psychoac.py:164
            const double fnum = S.binHz * (((p - 1) * x0 + p * x1) + (p + 1) * x2);
            double fm = EXACT ? fnum / s3 : fnum * recip_nr(s3);                  // psychoac.py:165
            // psychoac.py:27-29.  The fast path multiplies by the reciprocals of the constants 7500, 1000 and 10
            // (one rounding more each, against ~12 instructions per fp64 division) and uses atan_pos; EXACT divides and
            // calls atan like the reference
            double q = EXACT ? fm / 7500. : fm * (1. / 7500.);
            const double zm = EXACT ? 13 * atan(0.76 * fm / 1000.) + 3.5 * atan(q * q)
                                    : 13 * atan_pos((0.76 * fm) * 1e-3) + 3.5 * atan_pos(q * q);
            const double lvl15 = level - 15.0;                               // psychoac.py:42-43 (tonal drop)
            const double boost = 0.37 * fmax(level - 40, 0.0);               // psychoac.py:76
            double* e = mt + 4 * before;
            if (EXACT) {
                e[1] = zm;
                e[0] = lvl15;
                e[2] = boost;
            } else {
                // psychoac.py:14-18: 10^((spl-96)/10) as 2^(x log2 10), exponent in double-double (<= 1 ulp)
                const double xe = (lvl15 - 96) * 0.1;
                const double eh = xe * kLog2Of10;
                const double I = exp2_dd(eh, fma(xe, kLog2Of10, -eh) + xe * kLog2Of10Lo);
                const double ph = kLowHi * zm;
                const double pl = fma(kLowHi, zm, -ph) + kLowLo * zm;
                // (the entry leaves as two 16-byte stores: 8-byte stores 32 bytes apart from lane to lane meet on four banks)
                const double slope = (((-27 + boost) * 0.1) * kLog2Of10) * (double)TAB;  // upper slope, 1/T bit per Bark
                slLo = fmin(slLo, slope);
                slHi = fmax(slHi, slope);
                reinterpret_cast<double2*>(e)[0] = make_double2(I, zm);
                reinterpret_cast<double2*>(e)[1] = make_double2(slope, I * exp2_dd(ph, pl));
                // first line that sees this masker at all (fl(z_k - z_m) >= -1/2) and first line more than
                // 1/2 Bark above it (fl(z_k - z_m) > 1/2): both predicates are monotone in k
                // The searches start from the precomputed answers for the line nearest to the masker's own
                // frequency and walk to the exact boundary (a step or two; any start gives the same result).
                const int kNear = min(max((int)(fm * S.linesPerHz), 0), M - 1);
                MRC_PHASE(22);
                int lo = S.loLine[kNear], hi = S.hiLine[kNear];
#ifdef MRC_PROFILE_PHASES
                asm volatile("" : "+v"(lo), "+v"(hi));
#endif
                MRC_PHASE(23);
                // The hints are the answers for the Bark value of line kNear, less than a line away from z_m: the boundary
                // is the hinted line or a neighbour.  Both windows (hint - 2 .. hint + 1) are read at once and decided in
                // registers -- one LDS round trip instead of one per step of four dependent loops; whoever is not settled by
                // that (never, on the corpora of the tests) walks as before.
                {
                    auto zAt = [&](int k) { return zbS[min(max(k, 0), M - 1)]; };
                    const double a0 = zAt(lo - 2), a1 = zAt(lo - 1), a2 = zAt(lo), a3 = zAt(lo + 1);
                    const double b0 = zAt(hi - 2), b1 = zAt(hi - 1), b2 = zAt(hi), b3 = zAt(hi + 1);
                    // (line M stands for "no line": the predicate holds there; below line 0 it does not)
                    auto sees = [&](double zv, int k) { return k >= M || (k >= 0 && zv - zm >= -0.5); };
                    auto above = [&](double zv, int k) { return k >= M || (k >= 0 && zv - zm > 0.5); };
                    const bool s0 = sees(a0, lo - 2), s1 = sees(a1, lo - 1), s2 = sees(a2, lo), s3 = sees(a3, lo + 1);
                    const bool u0 = above(b0, hi - 2), u1 = above(b1, hi - 1), u2 = above(b2, hi), u3 = above(b3, hi + 1);
                    const int first = (s1 && !s0) ? lo - 1 : (s2 && !s1) ? lo : (s3 && !s2) ? lo + 1 : -1;
                    const int over = (u1 && !u0) ? hi - 1 : (u2 && !u1) ? hi : (u3 && !u2) ? hi + 1 : -1;
                    if (__any(first < 0 || over < 0)) {
                        while (lo > 0 && zbS[lo - 1] - zm >= -0.5) --lo;
                        while (lo < M && !(zbS[lo] - zm >= -0.5)) ++lo;
                        hi = max(hi, lo);
                        while (hi > 0 && zbS[hi - 1] - zm > 0.5) --hi;
                        while (hi < M && !(zbS[hi] - zm > 0.5)) ++hi;
                    } else {
                        lo = first;
                        hi = over;
                    }
                }
                atomicAdd(reinterpret_cast<unsigned int*>(cntArr) + (lo >> 1), 1u << (16 * (lo & 1)));
                atomicAdd(reinterpret_cast<unsigned int*>(nUpArr) + (hi >> 1), 1u << (16 * (hi & 1)));
                MRC_PHASE(24);
            }
        }
    }
    if (!EXACT) {
        slLo = -wave_max(-slLo);
        slHi = wave_max(slHi);
        if (lane == 0) {
            atomicMin(&slopeKey[0], order_key(slLo));
            atomicMax(&slopeKey[1], order_key(slHi));
        }
    }
    MRC_PHASE(4);
    __syncthreads();
    MRC_PHASE(12); MRC_STOP(4);

    // psychoac.py:214-217: SMR of a band = max over its lines of (SPL of the line - masked threshold),
    // accumulated with LDS integer max-atomics on an order-preserving key (initialised by the table
    // build's barrier below)
    const int scale = oscale[unit];
    const double* X = lines + (int64_t)unit * M;

    if (EXACT) {
        for (int base = 0; base < M; base += NT * kLinesPerThread) {
            double z[kLinesPerThread], tot[kLinesPerThread];
#pragma unroll
            for (int j = 0; j < kLinesPerThread; ++j) {
                int k = base + tid + j * NT;
                bool ok = k < M;
                z[j] = ok ? S.zb[k] : 0.0;
                tot[j] = ok ? S.quiet[k] : 0.0;
            }
            // psychoac.py:166-168 + 68-78: add every masker's spread intensity, in masker order
            for (int m = 0; m < nPeaks; ++m) {
                const double lvl = mt[4 * m], zm = mt[4 * m + 1], boost = mt[4 * m + 2];
#pragma unroll
                for (int j = 0; j < kLinesPerThread; ++j) {
                    double dz = z[j] - zm;
                    double adz = fabs(dz);
                    double t = adz - 0.5;
                    double arg = lvl;
                    if (adz > 0.5) arg = lvl + (-27 * t);
                    if (dz > 0.5) arg = arg + boost * t;
                    tot[j] += pow(10.0, (arg - 96) / 10);
                }
            }
#pragma unroll
            for (int j = 0; j < kLinesPerThread; ++j) {
                int k = base + tid + j * NT;
                if (k < M) {
                    double thr = spl_db(tot[j]);                                 // psychoac.py:173
                    if (thresh) thresh[(int64_t)unit * M + k] = thr;
                    double xs = ldexp(X[k], scale);                              // codecThem.py:323 (exact)
                    double spl = spl_db(2. * (xs * xs) / (1. / 2.)) - 6. * scale;   // psychoac.py:212
                    atomicMax(&bandKey[S.bandOfLine[k]], order_key(spl - thr));
                    if (wantPeak)
                        atomicMax(&peakKey[S.bandOfLine[k]], (unsigned long long)__double_as_longlong(fabs(X[k])));
                }
            }
        }
    } else {
        // suffix sums of the lower-side constants: sc[m] = sum_{j >= m} I_j 2^(b z_j), sc[nPeaks] = 0
        double* sc = xi;                                 // xi is dead (all peak reads happened before the barrier)
        if (TAB != kExpTab) smem[4 * H + kTabLongOff + tid] = e2Pre;      // (NT = 256 = TAB: an entry per thread)
        const int waveU = __builtin_amdgcn_readfirstlane(wave);          // (uniform: chunk indices stay in SGPRs)
        // ---- which evaluation of the upper-side sum the frame takes (wave-uniform): slope nodes (see kNodeR) when its
        // maskers are many and their slopes lie within reach of R nodes, else the sorted sweep.  Long blocks only.
        constexpr bool kNodes = (DIM == 1024 || DIM == 576) && NT == 256;      // (576: 156 of a transition block's <= 237 maskers)
        constexpr int kNodeMaxMaskers = kNodes ? node_max_maskers(DIM ? DIM : 1024) : 0;
        static_assert(!kNodes || kNodeMaxMaskers >= 128, "slope nodes: too few rows for this block shape");
        [[maybe_unused]] double nodeH = 0.0, nodeS0 = 0.0;               // node spacing / shallowest node (1/TAB bit per Bark)
        bool useNodes = false;
        if constexpr (kNodes) {
            const double lo = order_value(slopeKey[0]), hi = order_value(slopeKey[1]);
            nodeH = fmax((hi - lo) * (1.0 / (kNodeR - 1 - 2 * kNodeMargin)), kNodeHMin * TAB);
            nodeS0 = hi + kNodeMargin * nodeH;
            useNodes = nPeaks >= kNodeMinMaskers && nPeaks <= kNodeMaxMaskers && nodeH <= kNodeHMax * TAB;
#ifdef MRC_NODES_OFF
            useNodes = false;
#endif
        }
        // with nodes their rows take the place of the in-band prefix sums (and of the Bark grid before them), which move
        // behind the masker table
        double* const nodeQ = reinterpret_cast<double*>(pkBin);   // [<= 78][kNodeCols]: from the (dead) peak bins to the log10 table
        double* piH = piHi;
        double* piL = piLo;
        if (kNodes && useNodes) { piH = mt + 4 * nPeaks; piL = piH + (nPeaks + 1); }
        // kWave * kSeg >= the block's maximum number of peaks + 1: 512 >= N/4 in general; a block of DIM lines has at
        // most (DIM - 101) / 2 (13 for the short block: one per lane; 237 for the transition blocks: four; 461 for the long
        // block: eight -- but a frame that takes the slope nodes has at most 308: five.  Besides the shorter serial chain,
        // five entries of 32 bytes per lane put the lanes 160 bytes apart; at 256 bytes all 64 read the same bank)
        constexpr int kSegAny = DIM == 128 ? 1 : DIM == 576 ? 4 : 8;
        constexpr int kSegNodes = DIM == 1024 ? 5 : kSegAny;
        static_assert(DIM != 1024 || kWave * kSegNodes > node_max_maskers(1024), "segments of the scans");
        auto scan_sc = [&](auto segC) {
            constexpr int kSeg = decltype(segC)::value;
            double loc[kSeg];
            double run = 0.0;
            const int seg = kWave - 1 - lane;           // lanes take the segments in REVERSE order, so that the suffix
#pragma unroll                                          // over segments is a prefix over lanes (DPP shifts go up)
            for (int i = kSeg - 1; i >= 0; --i) {
                const int m = seg * kSeg + i;
                run += (m < nPeaks) ? mt[4 * m + 3] : 0.0;
                loc[i] = run;
            }
            const double incl = wave_incl_scan(run);    // inclusive prefix over lanes of the segment totals
            const double higher = dpp_shift_or_zero<0x138, 0xf>(incl);      // wave_shr:1 -> exclusive: the higher segments
#pragma unroll
            for (int i = 0; i < kSeg; ++i) {
                const int m = seg * kSeg + i;
                if (m < nPeaks) sc[m] = loc[i] + higher;
            }
            if (lane == 0) sc[nPeaks] = 0.0;
        };
        auto scan_pi = [&](auto segC) {
            // pi[m] = I_0 + ... + I_{m-1} in double-double: the in-band sum of a line is a DIFFERENCE of two
            // prefix sums, and with ~106 bits the difference is exact to far below one ulp of the result even
            // when a loud masker sits in the prefix (dynamic range of I within a frame < 2^50)
            constexpr int kSeg = decltype(segC)::value;
            double hi = 0.0, lo = 0.0, locH[kSeg], locL[kSeg];
#pragma unroll
            for (int i = 0; i < kSeg; ++i) {
                const int m = lane * kSeg + i;
                locH[i] = hi; locL[i] = lo;                            // exclusive within the segment
                dd_add(&hi, &lo, (m < nPeaks) ? mt[4 * m] : 0.0, 0.0);
            }
            double inH = hi, inL = lo;                                  // inclusive prefix scan of the segment totals
#define MRC_DD_SCAN_STEP(CTRL, MASK)                                                           \
            dd_add(&inH, &inL, dpp_shift_or_zero<CTRL, MASK>(inH), dpp_shift_or_zero<CTRL, MASK>(inL));
            MRC_DD_SCAN_STEP(0x111, 0xf) MRC_DD_SCAN_STEP(0x112, 0xf) MRC_DD_SCAN_STEP(0x114, 0xf)
            MRC_DD_SCAN_STEP(0x118, 0xf) MRC_DD_SCAN_STEP(0x142, 0xa) MRC_DD_SCAN_STEP(0x143, 0xc)
#undef MRC_DD_SCAN_STEP
            const double exH = dpp_shift_or_zero<0x138, 0xf>(inH), exL = dpp_shift_or_zero<0x138, 0xf>(inL);   // exclusive
#pragma unroll
            for (int i = 0; i < kSeg; ++i) {
                const int m = lane * kSeg + i;
                if (m <= nPeaks) {
                    double h = exH, l = exL;
                    dd_add(&h, &l, locH[i], locL[i]);
                    piH[m] = h; piL[m] = l;
                }
            }
        };
        auto scan_counts = [&](unsigned short* arr) {
            // per-line masker counts: inclusive prefix sums of the two histograms the table build left
            // (cnt[k] = maskers with fl(z_k - z_m) >= -1/2, nUp[k] = maskers with fl(z_k - z_m) > 1/2)
            const int per2 = (M + kWave) / kWave;                            // entries per lane, covers 0..M
            const int k0 = lane * per2, k1 = min(k0 + per2, M + 1);
            int sum = 0;
            for (int k = k0; k < k1; ++k) sum += arr[k];
            int run = wave_incl_scan(sum, lane) - sum;
            for (int k = k0; k < k1; ++k) {
                run += arr[k];
                arr[k] = (unsigned short)run;
            }
        };
        // ... of a 1024-line block, sixteen counts (eight words) per lane in registers: a prefix inside each word, the running
        // total added to both halves (counts <= 461 < 2^16), the lanes' totals scanned with DPP.  (Entry M -- maskers no line
        // sees -- is not read after the scan and stays as it is.)
        [[maybe_unused]] auto scan_counts_1024 = [&](unsigned short* arr) {
            unsigned* w = reinterpret_cast<unsigned*>(arr) + 8 * lane;
            unsigned x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = w[j];
            unsigned carry = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned v = x[j] + (x[j] << 16);
                v += carry * 0x10001u;
                x[j] = v;
                carry = v >> 16;
            }
            const unsigned before = (unsigned)(wave_incl_scan((int)carry, lane) - (int)carry) * 0x10001u;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = x[j] + before;
        };
        // The node terms, a masker per thread: lambda_r(theta) in product form (prefix x suffix products of theta - j), the two
        // 2^x, the R terms and the two error-bound terms; the four maskers of a row are lanes 4 j .. 4 j + 3 of a wave and
        // are summed there (two DPP steps); row j + 1 of Q gets the row's total, to be turned into prefix sums by node_scan.
        [[maybe_unused]] auto node_terms = [&]() {
            const double* tab64 = smem + 2 * H - kExpTab;                // 2^(j/64) (the 256-entry table is being staged)
            const double invH = 1.0 / nodeH;
            const double s0q = nodeS0 * ((double)kExpTab / TAB), hq = nodeH * ((double)kExpTab / TAB);   // 1/64 bit per Bark
            if (tid < kNodeCols) nodeQ[tid] = 0.0;                       // row 0: no masker below
            for (int base = waveU * kWave; base < nPeaks; base += NT) {  // (wave-uniform)
                const int m = base + lane;
                const bool valid = m < nPeaks;
                const int mm = min(m, nPeaks - 1);
                const double2 Iz = *reinterpret_cast<const double2*>(mt + 4 * mm);       // (one 16-byte read: see the table's stores)
                const double I = valid ? Iz.x : 0.0, zm = Iz.y, sl = mt[4 * mm + 2];
                const double theta = (nodeS0 - sl) * invH;              // the masker's slope in node units, [margin, R-1-margin]
                double suf[kNodeR];                                      // prod_{j > r} (theta - j)
                suf[kNodeR - 1] = 1.0;
#pragma unroll
                for (int r = kNodeR - 2; r >= 0; --r) suf[r] = suf[r + 1] * (theta - (r + 1));
                const double F0 = I * exp2_tab64<kExpTab>(-s0q, zm, tab64);      // I 2^(-sigma_0 z_m)
                const double gm = exp2_tab64<kExpTab>(hq, zm, tab64);            // 2^(h z_m)
                double G[kNodeCols];
                double F = F0, pre = 1.0, lsum = 0.0;
#pragma unroll
                for (int r = 0; r < kNodeR; ++r) {
                    const double lam = (pre * kNodeW.c[r]) * suf[r];    // lambda_r(theta)
                    G[r] = lam * F;
                    lsum += fabs(lam);
                    F *= gm;
                    pre *= theta - r;
                }
                G[kNodeR] = I * (fabs(pre) * kInvFactorial[kNodeR]);    // I |prod_r (theta - r)| / R!
                G[kNodeR + 1] = lsum * F0;
                const bool store = (lane & 3) == 0 && valid;            // (the row's first masker exists)
                double* rowOut = nodeQ + ((m >> 2) + 1) * kNodeCols;
#pragma unroll
                for (int j = 0; j < kNodeCols; ++j) {
                    double v = G[j];
                    v += dpp_move<0xB1>(v);                              // quad_perm [1,0,3,2]
                    v += dpp_move<0x4E>(v);                              // quad_perm [2,3,0,1]
                    if (store) rowOut[j] = v;
                }
            }
        };
        // Row totals -> prefix sums, in place, by two waves (nine columns each): lane = (seventh of the rows, column); a lane loads
        // its <= 11 rows of the column at once and sums them up in registers; the sevenths of a column get the totals below them
        // by a shift and a three-step scan through ds_bpermute (lane - 9 d holds the same column, d sevenths lower).
        [[maybe_unused]] auto node_scan = [&](int half) {
            static_assert(kNodeCols == 18 && kNodeScanSegs * 9 <= kWave, "columns of the row scan");
            static_assert((node_max_maskers(1024) + kNodeC - 1) / kNodeC <= kNodeScanSegs * kNodeSeg, "rows of the row scan");
            const int nR = (nPeaks + kNodeC - 1) / kNodeC;               // rows 1 .. nR hold totals; row q becomes sum_{m < 4 q}
            const int L = (nR + kNodeScanSegs - 1) / kNodeScanSegs;      // <= kNodeSeg
            const int seg = (lane * 57) >> 9;                            // lane / 9 for lane < 64
            const int col = 9 * half + (lane - 9 * seg);
            const bool live = seg < kNodeScanSegs;
            double v[kNodeSeg];
#pragma unroll
            for (int i = 0; i < kNodeSeg; ++i) {
                const int r = 1 + seg * L + i;
                v[i] = (live && i < L && r <= nR) ? nodeQ[r * kNodeCols + col] : 0.0;
            }
#pragma unroll
            for (int i = 1; i < kNodeSeg; ++i) v[i] += v[i - 1];
            const double tot = v[kNodeSeg - 1];
            // the totals BELOW a seventh: an inclusive scan of the totals shifted up by one seventh.  (Not "inclusive minus
            // own": the rows grow by 2^6 .. 2^9 per Bark, and the small sum of the lower rows would be lost in the subtraction.)
            auto from_below = [&](double x, int d) {
                const int from = 4 * (lane - 9 * d);
                const double y = __hiloint2double(__builtin_amdgcn_ds_bpermute(from, __double2hiint(x)),
                                                  __builtin_amdgcn_ds_bpermute(from, __double2loint(x)));
                return seg >= d ? y : 0.0;
            };
            double off = from_below(tot, 1);
#pragma unroll
            for (int d = 1; d < kNodeScanSegs; d *= 2) off += from_below(off, d);
#pragma unroll
            for (int i = 0; i < kNodeSeg; ++i) {
                const int r = 1 + seg * L + i;
                if (live && i < L && r <= nR) nodeQ[r * kNodeCols + col] = v[i] + off;
            }
        };
        bool scansDone = false;
        if constexpr (kNodes) {
            if (useNodes) {                              // (workgroup-uniform)
                // The node terms by every wave.  With more than NT maskers (more than half of the frames of noise) wave 0 builds
                // the terms of the rest in a second round: the scans that need the masker table only then follow on waves
                // 1 .. 3 without a barrier, and the scan over the rows, which needs every wave's terms, comes behind the barrier
                // on two waves.  Otherwise all scans run side by side behind the barrier.  (What a barrier-delimited phase
                // costs is its LONGEST wave: the others hold their slots idle.)
                MRC_PHASE(13);
                if (!(MRC_PROFILE_NODESKIP & 1)) node_terms();
                MRC_PHASE(14);
                using SegN = std::integral_constant<int, kSegNodes>;
                auto counts = [&](unsigned short* arr) {
                    if constexpr (DIM == 1024) scan_counts_1024(arr); else scan_counts(arr);
                };
#ifndef MRC_SCAN_HYBRID
#define MRC_SCAN_HYBRID 1
#endif
                if (!MRC_SCAN_HYBRID || nPeaks > NT) {   // (workgroup-uniform) wave 0 has had a second round of terms
                    if (waveU == 1) scan_pi(SegN{});
                    else if (waveU == 2) { scan_sc(SegN{}); counts(cntArr); }
                    else if (waveU == 3) counts(nUpArr);
                    MRC_PHASE(9);                        // (profiling build: the slot of the sorted sweep's near field)
                    __syncthreads();
                    MRC_PHASE(15);
                    if (!(MRC_PROFILE_NODESKIP & 1) && (waveU == 2 || waveU == 3)) node_scan(waveU - 2);
                } else {                                 // every wave is through with its terms at the same time
                    __syncthreads();
                    MRC_PHASE(15);
                    if (waveU < 2) { if (!(MRC_PROFILE_NODESKIP & 1)) node_scan(waveU); }
                    else if (waveU == 2) scan_pi(SegN{});
                    else { scan_sc(SegN{}); counts(cntArr); counts(nUpArr); }
                    MRC_PHASE(9);
                }
                scansDone = true;
            }
        }
        if (!scansDone) {
            // four independent scans, dealt to the workgroup's waves (4 waves: one each; 2 waves: two each)
            for (int task = waveU; task < 4; task += NT / kWave) {
                using SegA = std::integral_constant<int, kSegAny>;
                if (task == 0) scan_sc(SegA{});
                else if (task == 1) scan_pi(SegA{});
                else if constexpr (DIM == 1024) scan_counts_1024(task == 2 ? cntArr : nUpArr);
                else scan_counts(task == 2 ? cntArr : nUpArr);
            }
        }
        __syncthreads();
        MRC_PHASE(5); MRC_STOP(5);

        // Each wave sweeps 64-line chunks (one line per lane); the chunk order pairs cheap (low) with
        // expensive (high) chunks so the four waves finish together.  Per line, the Bark-sorted maskers
        // split into [0, nUp): more than 1/2 Bark below the line (upper slope, needs 2^x),
        // [nUp, cnt): within +-1/2 Bark (contributes exactly I_m), [cnt, P): more than 1/2 Bark above
        // (lower slope, served by the suffix sums).
        const int nChunks = (M + kWave - 1) / kWave;
        const int nWaves = NT / kWave;
        // the per-line constants of the NEXT chunk are loaded while this one is computed (loop-carried, so the
        // global-load latency is never exposed between the loops of a chunk)
        struct LineConst { double z, quiet, lowE, x; int bnd; };
        auto chunk_of = [&](int i) { return i * nWaves + ((i & 1) ? (nWaves - 1 - waveU) : waveU); };
        auto load_consts = [&](int i) {
            // (byte offsets as 32-bit unsigned values: scalar base + vector offset addressing, no 64-bit address arithmetic)
            const unsigned kc = (unsigned)min(chunk_of(i) * kWave + lane, M - 1);
            const char* lc = reinterpret_cast<const char*>(S.lineC) + kc * (unsigned)sizeof(LineConstants);
            const double2 a = *reinterpret_cast<const double2*>(lc);
            const double2 b = *reinterpret_cast<const double2*>(lc + 16);
            const double x = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(X) + kc * 8u);
            return LineConst{a.x, a.y, b.x, x, __double2loint(b.y)};
        };
        // in-band maskers [from, cnt) and the lower side on top of `tot` (quiet threshold + upper side): the line's masked intensity
        auto tail_sum = [&](double tot, int cnt, int from, double lowE) {
            if (cnt > from) {
                // sum of I_m over [from, cnt) = pi[cnt] - pi[from], in double-double
                const double ah = piH[cnt], al = piL[cnt], bh = piH[from], bl = piL[from];
                const double d1 = ah - bh;
                const double v = d1 - ah;
                const double e = ((ah - (d1 - v)) - (bh + v)) + (al - bl);
                tot += d1 + e;
            }
            // maskers more than 1/2 Bark above the line: -27 dB/Bark for all of them
            return fma(lowE, sc[cnt], tot);
        };
        // psychoac.py:173,212-217: SMR of a band = max over its lines of SPL(4 xs^2) - 6 scale - SPL(t).  Unless one
        // of the two SPLs sits on its -30 dB floor (digital silence) that is 10 log10(4 xs^2 / t) - 6 scale, and
        // log10 is monotone: the band maximum of the RATIO is taken and converted once per band at the end
        // instead of two log10 per line (the difference to the reference's order of roundings is ~1e-14 dB, five
        // orders below what the FFT in front of it already differs by).  Lines on the floor, and every line when
        // the caller wants the thresholds themselves, take the reference's formula.
        // (a2 of a line: the intensity of its own MDCT line, psychoac.py:212)
        // = 2 xs^2 / (1/2) with xs = x 2^scale (codecThem.py:323; psychoac.py:212): the factors of two commute with the one
        // rounding of the square, so the square of 2 xs is the same double
        auto line_a2 = [&](const LineConst& cur) {
            const double xs2 = ldexp(cur.x, scale + 1);
            return xs2 * xs2;
        };
        auto line_plain = [&](double a2, double t) { return thresh != nullptr || !(a2 >= kSplFloorGuard && t >= kSplFloorGuard); };
        // noPlain: the caller has checked that no lane of the chunk takes the reference's formula (no call in its loop)
        auto finish = [&](const LineConst& cur, int k, double t, auto noPlain) {
            const double a2 = line_a2(cur);
            const bool plain = decltype(noPlain)::value ? false : line_plain(a2, t);
            double ex = -1e300, q = 0.0;
            if (plain) {
                double thr;
                ex = excess_plain(t, a2, scale, logTab, &thr);
                if (thresh && k < M) thresh[(int64_t)unit * M + k] = thr;
            } else {
                q = a2 * recip_nr(t);
            }
            const int bnd = cur.bnd;                     // lanes past the end repeat the last line: maxima unchanged
            if (__all(bnd == __builtin_amdgcn_readfirstlane(bnd))) {
                // whole chunk inside one band (the wide top bands): 64 lanes on one LDS address would be served one by
                // one; the maximum of each 16-lane row is taken in registers and four lanes go to the LDS
                const bool rowHead = (lane & 15) == 0;
                const double qBest = row_max(q);
                if (rowHead) atomicMax(&ratioKey[bnd], (unsigned long long)__double_as_longlong(qBest));
                if (__any(plain)) {
                    const double best = row_max(ex);
                    if (rowHead) atomicMax(&bandKey[bnd], order_key(best));
                }
                if (wantPeak) {
                    const double pk = row_max(fabs(cur.x));
                    if (rowHead) atomicMax(&peakKey[bnd], (unsigned long long)__double_as_longlong(pk));
                }
            } else {
                atomicMax(&ratioKey[bnd], (unsigned long long)__double_as_longlong(q));
                if (plain) atomicMax(&bandKey[bnd], order_key(ex));
                if (wantPeak) atomicMax(&peakKey[bnd], (unsigned long long)__double_as_longlong(fabs(cur.x)));
            }
        };
        __builtin_amdgcn_s_setprio(0);
        const double slMid = 0.5 * (order_value(slopeKey[0]) + order_value(slopeKey[1]));
        const double spreadHalf = 0.5 * (order_value(slopeKey[1]) - order_value(slopeKey[0])) * (0.6931471805599453094 / TAB);
#if defined(MRC_PROFILE_NOSWEEP)                  // profiling aids (wrong results): no unit / units with (unit & n) skip the sweep
        const bool sweepOn = false;
#elif defined(MRC_PROFILE_HALFSWEEP)
        const bool sweepOn = !(unit & MRC_PROFILE_HALFSWEEP);
#else
        const bool sweepOn = true;
#endif
        if (kNodes && useNodes) {
            if constexpr (kNodes) {
            // ---- slope nodes: per line two 2^x, a Horner pass over its row of prefix sums, and the direct pairs of the
            // maskers between the row and nUp
            MRC_NODE_COUNT(0);
            const double xr = (nodeH * kNodeR) / (-nodeS0);
            double ps = xr * xr;
            ps *= ps; ps *= ps; ps *= ps;                // (h R / |sigma_0|)^16
            const double psiStar = ps * kExpMinus16;
            // The chunks where the evaluation below is not the last word -- a line whose error bound fails, a line on the
            // SPL floor, every chunk when the caller wants the thresholds -- are set aside (a bit per chunk of the wave) and
            // done after the loop: the out-of-line calls they need would otherwise sit in the hot loop and cost it the
            // scalar registers a call clobbers (its pointers and masks were being reloaded from a spill lane every chunk).
            struct NodeEval { double t, bound; int cnt, nUp; };
            auto node_chunk = [&](const LineConst& cur, int kc) {
                const int cnt = cntArr[kc], nUp = nUpArr[kc];      // maskers that reach the line / lie > 1/2 Bark below it
                const double zq = cur.z - 0.5;
                const int q = nUp >> 2, rem = nUp & 3;   // (kNodeC = 4)
                const double* row = nodeQ + q * kNodeCols;
                const double E0 = exp2_tab64<TAB>(nodeS0, zq, e2tab);
                const double g = exp2_tab64<TAB>(-nodeH, zq, e2tab);
                const double errBound = fma(psiStar, row[kNodeR], (kNodeRoundEps * E0) * row[kNodeR + 1]);
                const double up = (MRC_PROFILE_NODESKIP & 2) ? node_line<0, TAB>(row, mt, e2tab, 4 * q, rem, nPeaks - 1, zq, E0, g)
                                                             : node_line<kNodeC - 1, TAB>(row, mt, e2tab, 4 * q, rem, nPeaks - 1, zq, E0, g);
                return NodeEval{tail_sum(cur.quiet + up, cnt, nUp, cur.lowE), errBound, cnt, nUp};
            };
            unsigned setAside = 0;                       // (wave-uniform)
            LineConst nxt = load_consts(0);
            for (int i = 0; sweepOn && chunk_of(i) < nChunks; ++i) {
                const int k = chunk_of(i) * kWave + lane;
                const LineConst cur = nxt;
                nxt = load_consts(i + 1);
                if (haveSwitch && !__any(needBand[cur.bnd])) continue;                           // (see needBand)
                MRC_PHASE(6);
                const NodeEval ev = node_chunk(cur, min(k, M - 1));
                MRC_PHASE(8);
                const bool odd = (!MRC_PROFILE_NODESKIP && !(ev.bound <= kNodeTol * ev.t)) || line_plain(line_a2(cur), ev.t);
                if (__any(odd)) { setAside |= 1u << i; continue; }
                MRC_NODE_COUNT(2);
                finish(cur, k, ev.t, std::true_type{});
                MRC_PHASE(10);
            }
            while (setAside) {
                const int i = __builtin_ctz(setAside);
                setAside &= setAside - 1;
                const int c = chunk_of(i);
                const int k = c * kWave + lane;
                const LineConst cur = load_consts(i);
                const NodeEval ev = node_chunk(cur, min(k, M - 1));
                double t = ev.t;
                if (!MRC_PROFILE_NODESKIP && __any(!(ev.bound <= kNodeTol * ev.t))) {
                    // a line of this chunk lives on what the interpolation does worst: the chunk goes back to the sorted sweep
                    MRC_NODE_COUNT(3);
                    if (sens && lane == 0) atomicAdd(&sens[4], 1ull);
                    const double tot = cur.quiet + upper_cold<TAB>(mt, e2tab, S.zb, M, c, lane, ev.nUp, ev.cnt, cur.z, slMid, spreadHalf);
                    t = tail_sum(tot, ev.cnt, __builtin_amdgcn_readlane(ev.nUp, kWave - 1), cur.lowE);
                } else {
                    MRC_NODE_COUNT(2);
                }
                finish(cur, k, t, std::false_type{});
            }
            }
        } else {
        // ---- sorted sweep.  Rounds of up to four chunks per wave.  Pass 1 evaluates the FAR FIELD of the round's chunks --
        // the only part that needs a large register tile (the expansion coefficients) -- and keeps one value per line; pass 2
        // does the near maskers, the in-band and lower-side sums and the SPL conversions with that value added in.
        MRC_NODE_COUNT(1);
        for (int i0 = 0; sweepOn && chunk_of(i0) < nChunks; i0 += 4) {
        __builtin_amdgcn_s_setprio(MRC_FAR_PRIO);
        double far0 = 0.0, far1 = 0.0, far2 = 0.0, far3 = 0.0;
        unsigned farMask = 0;                            // bit u: chunk u of the round took the far field
        // (a block of DIM lines has at most (DIM - 101) / 2 maskers: a short block's 13 never reach kFarMinMaskers, so its
        // instance carries no far-field code -- and fits the registers of eight waves per SIMD)
        constexpr bool kHaveFar = DIM == 0 || (DIM - 101) / 2 >= kFarMinMaskers;
        for (int u = 0; kHaveFar && u < 4; ++u) {
            const int c = chunk_of(i0 + u);
            if (c >= nChunks) break;
            const int kc = min(c * kWave + lane, M - 1);
            if (haveSwitch && !__any(needBand[S.bandOfLine[kc]])) continue;                      // (see needBand)
            const int nFar = __builtin_amdgcn_readfirstlane((int)nUpArr[kc]);      // nUp of the chunk's first line
            double acc = 0.0;
            if (!far_eval<TAB, kHaveFar>(mt, e2tab, S.zb, M, c, lane, nFar, S.zb[kc], slMid, spreadHalf, &acc)) continue;
            farMask |= 1u << u;
            far0 = u == 0 ? acc : far0;
            far1 = u == 1 ? acc : far1;
            far2 = u == 2 ? acc : far2;
            far3 = u == 3 ? acc : far3;
        }
        MRC_PHASE(7);
        // ---- pass 2
        __builtin_amdgcn_s_setprio(0);
        LineConst nxt = load_consts(i0);
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            const int c = chunk_of(i);
            if (c >= nChunks) break;
            const int k = c * kWave + lane;
            const int kc = min(k, M - 1);
            const LineConst cur = nxt;
            nxt = load_consts(i + 1);
            if (haveSwitch && !__any(needBand[cur.bnd])) continue;                               // (see needBand)
            // quiet threshold + far field (psychoac.py:155,166-168; the order of the additions is free, see above)
            double tot = cur.quiet + (u == 0 ? far0 : u == 1 ? far1 : u == 2 ? far2 : far3);
            const int cnt = cntArr[kc], nUp = nUpArr[kc];      // maskers that reach the line / lie > 1/2 Bark below it
            MRC_PHASE(6);
            tot = near_eval<TAB>(mt, e2tab, nUp, cnt, cur.z - 0.5, ((farMask >> u) & 1u) != 0, tot);
            MRC_PHASE(9);
            if (MRC_PROFILE_SKIP & 8) {
                if (tot + cur.lowE + cur.x == 12345.0 && cur.bnd == 77) bandKey[0] = 1;      // keep the loads alive
                continue;
            }
            // no line of the chunk is above the band of the maskers from max nUp on: a line that sees one is inside +-1/2 Bark
            finish(cur, k, tail_sum(tot, cnt, __builtin_amdgcn_readlane(nUp, kWave - 1), cur.lowE), std::false_type{});
            MRC_PHASE(10);
        }
        }
        }
    }
    __syncthreads();
    MRC_PHASE(11);
#ifdef MRC_PROFILE_PHASES
    __syncthreads();
    if (tid < 32 && (blockIdx.x & 63) == 0) atomicAdd(&gPhaseCycles[tid], tid == 31 ? 1ull : sPhase_[tid]);   // [31]: workgroups sampled
#endif
    for (int bnd = tid; bnd < S.nBands; bnd += NT) {
        double v = bandKey[bnd] ? order_value(bandKey[bnd]) : -1e300;              // lines on the SPL floor / EXACT
        if (!EXACT && ratioKey[bnd]) {
            const double q = __longlong_as_double((long long)ratioKey[bnd]);
            v = fmax(v, 10 * log10_tab32(q, logTab) - 6. * scale);
        }
        smr[(int64_t)unit * S.nBands + bnd] = v;
        // max |X| per band of the UNSCALED lines: what the scale factors need (codecThem.py:346), so the back end
        // does not have to read the lines once more for it
        if (wantPeak) bandPeak[(int64_t)unit * S.nBands + bnd] = __longlong_as_double((long long)peakKey[bnd]);
    }
}

template <bool EXACT, class SampleT, int NT, int DIM, int MODE>
__global__ __launch_bounds__(NT) MRC_SMR_OCC void smr_kernel(DevShape S, int nsigArg, const SampleT* __restrict__ chL,
                                                       const SampleT* __restrict__ chR, int64_t stride,
                                                       const int64_t* __restrict__ offsetsArg,
                                                       const double* __restrict__ lines,
                                                       const int* __restrict__ oscale, double* __restrict__ smr,
                                                       double* __restrict__ threshArg, double* __restrict__ bandPeakArg,
                                                       const int* __restrict__ msSwitch, SmrLds layArg,
                                                       unsigned long long* __restrict__ sens) {
    smr_body<EXACT, SampleT, NT, DIM, MODE>(S, nsigArg, chL, chR, stride, offsetsArg, lines, oscale, smr, threshArg,
                                            bandPeakArg, msSwitch, layArg, sens);
}

// ------------------------------------------------------------------------------------------------
// smr_short_kernel -- the same quantities as smr_kernel for the reference's SHORT block (a = b = 128: 128 lines, 28 searched
// bins, at most 13 maskers), one WAVEFRONT per unit and no workgroup barrier.  smr_kernel's machinery -- masker-side
// searches, histograms and scans, prefix / suffix sums, the sorted sweep with its far field -- pays for itself with hundreds
// of maskers; with thirteen it is overhead (1 800 VALU instructions per unit in two waves, 61 % VALU busy: `profiles/
// r03_shapes_sq_counters.txt`).  Here a lane owns two lines and adds the maskers one by one, in the reference's own order
// (psychoac.py:166-168): I_m inside +-1/2 Bark, I_m 2^(slope x (|dz| - 1/2)) outside, with the level-dependent slope above the
// masker and -27 dB/Bark below it -- one table-driven 2^x per (masker, line).  Maskers, SPL conversions, the ratio form of the
// band maximum and the floor handling are smr_kernel's (same helpers).  A wave walks `run` consecutive units; the tables
// (FFT twiddle quadrant, 2^x, log10) are staged once per workgroup, Hann values and per-line constants live in registers.
// ------------------------------------------------------------------------------------------------
// One masker's table entry {I, z, upper slope in 1/64 bit per Bark} from the sum of its three bins and the numerator of its
// centre frequency.  OUT OF LINE on purpose: it runs once per unit on at most 13 lanes, and inlined its ~60 polynomial
// constants would sit in registers across the whole unit loop (145 registers instead of ~100: a wave per SIMD less).
__device__ __attribute__((noinline)) void short_masker(double s3, double fnum, const double* logTab, double* e) {
    const double level = spl_db_tab(s3, logTab);                              // psychoac.py:164
    const double fm = fnum * recip_nr(s3);                                    // psychoac.py:165
    const double q = fm * (1. / 7500.);                                       // psychoac.py:27-29
    const double zm = 13 * atan_pos((0.76 * fm) * 1e-3) + 3.5 * atan_pos(q * q);
    const double lvl15 = level - 15.0;                                        // psychoac.py:42-43 (tonal drop)
    const double boost = 0.37 * fmax(level - 40, 0.0);                        // psychoac.py:76
    const double xe = (lvl15 - 96) * 0.1;                                     // psychoac.py:14-18
    const double eh = xe * kLog2Of10;
    e[0] = exp2_dd(eh, fma(xe, kLog2Of10, -eh) + xe * kLog2Of10Lo);
    e[1] = zm;
    e[2] = (((-27 + boost) * 0.1) * kLog2Of10) * (double)kExpTab;
}

// ... and the conversion of a band's maximal ratio (once per band and unit; out of line for the same reason)
__device__ __attribute__((noinline)) double short_band_db(double q, int scale, const double* logTab) {
    return 10 * log10_tab32(q, logTab) - 6. * scale;
}

constexpr int kShortWaves = 4;
constexpr int kShortWaveLds = 256 + 256 + 32 + 64 + 96;        // doubles per wave: A | B (FFT) | xi | masker table | three key arrays
constexpr int kShortSharedLds = 64 + kExpTab + kLogTabEntries * 4;   // twiddle quadrant | 2^(j/64) | log10 table
template <class SampleT, int MODE>
#ifndef MRC_SMR_SHORT_OCC
#define MRC_SMR_SHORT_OCC 4
#endif
__global__ __launch_bounds__(kWave * kShortWaves) __attribute__((amdgpu_waves_per_eu(MRC_SMR_SHORT_OCC, MRC_SMR_SHORT_OCC))) void smr_short_kernel(
    DevShape S, int64_t nUnits, int run, const SampleT* __restrict__ chL, const SampleT* __restrict__ chR, int64_t stride,
    const int64_t* __restrict__ offsets, const double* __restrict__ lines, const int* __restrict__ oscale,
    double* __restrict__ smr, double* __restrict__ bandPeak, const int* __restrict__ msSwitch) {
    constexpr int H = 128, M = 128, last = 28;
    constexpr int nsig = MODE == 1 ? 1 : 4;
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* ws = smem + wave * kShortWaveLds;
    double2* A = reinterpret_cast<double2*>(ws);
    double2* B = A + H;
    double* xi = ws + 4 * H;
    double* mt = xi + 32;                               // [<= 13][4]: I, z, upper slope (1/64 bit per Bark), -
    unsigned long long* ratioKey = reinterpret_cast<unsigned long long*>(mt + 64);
    unsigned long long* bandKey = ratioKey + 32;
    unsigned long long* peakKey = bandKey + 32;
    double* shared = smem + kShortWaves * kShortWaveLds;
    double2* Wq = reinterpret_cast<double2*>(shared);   // [32] first quadrant of e^{-2 pi i t/128}
    double* e2tab = shared + 64;
    double* logTab = e2tab + kExpTab;
    {
        const int t = threadIdx.x;
        if (t < H / 4) Wq[t] = S.wH[t];
        if (t < kExpTab) e2tab[t] = kExp2Tab[t];
        if (t < kLogTabEntries * 4) logTab[t] = kLogTabDev.v[t];
    }
    // lane constants: Hann values of the lane's two (even, odd) sample pairs, the split twiddle of bin `lane`, and of the
    // lane's two lines (k = lane, lane + 64): Bark value, quiet threshold, band
    double he[2], ho[2], zk[2], qk[2];
    int bk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = lane + kWave * j;
        he[j] = S.hann[2 * n]; ho[j] = S.hann[2 * n + 1];
        zk[j] = S.zb[n]; qk[j] = S.quiet[n]; bk[j] = S.bandOfLine[n];
    }
    const double2 wn = S.wN[min(lane, last - 1)];
    const double xiInv = 1.0 / S.xiDen;
    const TwQuarter W{Wq, H / 4 - 1, 5};
    const int nb = S.nBands;
    __syncthreads();                                    // tables visible (the only workgroup barrier)

    const int64_t first = ((int64_t)blockIdx.x * kShortWaves + wave) * run;
    for (int it = 0; it < run; ++it) {
        const int64_t unit = first + it;
        if (unit >= nUnits) break;                      // wave-uniform
        const int64_t f = unit / nsig;
        const int sig = (int)(unit % nsig);
        if (MODE == 2) {                                // a unit none of whose bands the switch selects: see smr_kernel
            const bool need = lane < nb && ((sig >= 2) == (msSwitch[f * nb + lane] != 0));
            if (!__any(need)) continue;
        }
        const int64_t off = offsets ? offsets[f] : f * stride;
        const bool pairAligned = !(off & 1) && !(reinterpret_cast<uintptr_t>(chL) & (2 * sizeof(SampleT) - 1)) &&
                                 (!chR || !(reinterpret_cast<uintptr_t>(chR) & (2 * sizeof(SampleT) - 1)));
        // Hann window (window.py:28-45), real FFT through a 128-point complex FFT
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = lane + kWave * j;
            const double2 eo = load_signal_pair(chL, chR, off + 2 * n, sig, pairAligned);
            A[n] = make_double2(eo.x * he[j], eo.y * ho[j]);
        }
        if (lane < 32) { ratioKey[lane] = 0ull; bandKey[lane] = 0ull; peakKey[lane] = 0ull; }
        const int scale = oscale[unit];
        const double* X = lines + unit * M;
        const double x0 = X[lane], x1 = X[lane + kWave];        // (in flight under the FFT)
        wave_sync_lds();
        fft_pass<4, true, TwQuarter, kWave>(A, B, H, 1, W, lane);
        wave_sync_lds();
        fft_pass<4, true, TwQuarter, kWave>(B, A, H, 4, W, lane);
        wave_sync_lds();
        fft_pass<4, true, TwQuarter, kWave>(A, B, H, 16, W, lane);
        wave_sync_lds();
        fft_pass<2, true, TwQuarter, kWave>(B, A, H, 64, W, lane);
        wave_sync_lds();
        if (lane < last) {                               // psychoac.py:147-151: intensity of bins 0 .. 27
            const int k = lane;
            const double2 zz = A[k];
            double2 zc = A[(H - k) % H];
            zc.y = -zc.y;
            const double2 ev = make_double2(0.5 * (zz.x + zc.x), 0.5 * (zz.y + zc.y));
            const double2 d = make_double2(zz.x - zc.x, zz.y - zc.y);
            const double2 od = make_double2(0.5 * d.y, -0.5 * d.x);
            double2 Xk = cmul(wn, od);
            Xk.x += ev.x; Xk.y += ev.y;
            xi[k] = (4. * (Xk.x * Xk.x + Xk.y * Xk.y)) * xiInv;
        }
        wave_sync_lds();
        // tonal maskers: strict 3-point peaks at bins 1 .. 26, in increasing bin order (psychoac.py:160-165)
        const int p = lane;
        double y0 = 0.0, y1 = 0.0, y2 = 0.0;
        if (p >= 1 && p <= last - 2) { y0 = xi[p - 1]; y1 = xi[p]; y2 = xi[p + 1]; }
        const bool isPeak = p >= 1 && p <= last - 2 && y1 > y0 && y1 > y2;
        const unsigned long long peaks = __ballot(isPeak);
        const int nPeaks = __popcll(peaks);
        if (isPeak) {
            const int idx = __popcll(peaks & ((1ull << lane) - 1ull));
            const double s3 = (y0 + y1) + y2;
            const double fnum = S.binHz * (((p - 1) * y0 + p * y1) + (p + 1) * y2);
            short_masker(s3, fnum, logTab, mt + 4 * idx);
        }
        wave_sync_lds();
        // psychoac.py:155,166-173: quiet threshold + every masker's spread intensity, in masker order
        double tot0 = qk[0], tot1 = qk[1];
        for (int m = 0; m < nPeaks; ++m) {
            const double I = mt[4 * m], zm = mt[4 * m + 1], sl = mt[4 * m + 2];
            const double d0 = zk[0] - zm, d1 = zk[1] - zm;
            const double u0 = fmax(fabs(d0) - 0.5, 0.0), u1 = fmax(fabs(d1) - 0.5, 0.0);
            tot0 = fma(I, exp2_tab64<kExpTab>(d0 > 0.0 ? sl : kLowHi * (double)kExpTab, u0, e2tab), tot0);
            tot1 = fma(I, exp2_tab64<kExpTab>(d1 > 0.0 ? sl : kLowHi * (double)kExpTab, u1, e2tab), tot1);
        }
        // psychoac.py:212-217 as in smr_kernel: band maximum of the intensity / threshold ratio, one log10 per band; lines on
        // the SPL floor take the reference's formula
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double x = j ? x1 : x0, t = j ? tot1 : tot0;
            const int bnd = bk[j];
            const double xs = ldexp(x, scale);                                    // codecThem.py:323 (exact)
            const double a2 = 2. * (xs * xs) / (1. / 2.);
            if (!(a2 >= kSplFloorGuard && t >= kSplFloorGuard)) {
                double thr;
                atomicMax(&bandKey[bnd], order_key(excess_plain(t, a2, scale, logTab, &thr)));
            } else {
                atomicMax(&ratioKey[bnd], (unsigned long long)__double_as_longlong(a2 * recip_nr(t)));
            }
            atomicMax(&peakKey[bnd], (unsigned long long)__double_as_longlong(fabs(x)));
        }
        wave_sync_lds();
        if (lane < nb) {
            double v = bandKey[lane] ? order_value(bandKey[lane]) : -1e300;
            if (ratioKey[lane]) v = fmax(v, short_band_db(__longlong_as_double((long long)ratioKey[lane]), scale, logTab));
            smr[unit * nb + lane] = v;
            bandPeak[unit * nb + lane] = __longlong_as_double((long long)peakKey[lane]);
        }
        wave_sync_lds();                                // keys read before the next unit clears them
    }
}

}  // namespace

// Two translation units.  The mono long-block kernel of the hot path -- smr_kernel<false, *, 256, 1024, 1>, 80 % of a
// headline step -- gains 3 % from LLVM's max-ILP scheduling strategy; the joint variant <..., 2> LOSES 2 % with it (more
// spills at the 128-register cap).  The strategy is a per-file option, so mrc_kernels_smr_mono.hip includes this file with
// MRC_SMR_TU_MONO defined and compiles that one instantiation (launch_smr_mono_long); everything else stays here.
// Diagnostics builds keep one unit: their counters are device globals of the unit that defines them.
#if defined(MRC_NODE_STATS) || defined(MRC_PROFILE_PHASES)
#define MRC_SMR_SPLIT 0
#else
#define MRC_SMR_SPLIT 1
#endif
#ifndef MRC_SMR_THREADS                          // workgroup size for blocks of more than 128 lines
#define MRC_SMR_THREADS 256
#endif
hipError_t launch_smr_mono_long(const DevShape& S, int64_t nFrames, const void* chL, int fmt, int64_t stride,
                                const int64_t* offsets, const double* lines, const int* oscale, double* smr,
                                double* bandPeak, hipStream_t st, unsigned long long* sens);

namespace {
// dynamic LDS (doubles): FFT ping-pong [4H] + intensity spectrum [peakLast + 1].  The staged tables go into
// areas that are dead when they are needed if there is room (the long block is sized for 4 workgroups per CU
// and must not grow), else behind the spectrum.
inline SmrLds smr_launch_layout(const DevShape& S, size_t* ldsBytes) {
    int total = 0;
    const SmrLds lay = smr_layout(S.H, S.halfN, S.peakLast, &total);
#ifdef MRC_PROFILE_EXTRA_LDS                     // profiling aid: pad the workgroup's LDS (occupancy experiments)
    total += MRC_PROFILE_EXTRA_LDS / 8;
#endif
    *ldsBytes = (size_t)total * sizeof(double);
    return lay;
}
}  // namespace

#ifdef MRC_SMR_TU_MONO
#if MRC_SMR_SPLIT
hipError_t launch_smr_mono_long(const DevShape& S, int64_t nFrames, const void* chL, int fmt, int64_t stride,
                                const int64_t* offsets, const double* lines, const int* oscale, double* smr,
                                double* bandPeak, hipStream_t st, unsigned long long* sens) {
    size_t lds = 0;
    const SmrLds lay = smr_launch_layout(S, &lds);
    const dim3 grid((unsigned)nFrames);
    if (fmt == kSampleI16)
        hipLaunchKernelGGL((smr_kernel<false, short, 256, 1024, 1>), grid, dim3(256), lds, st, S, 1, (const short*)chL,
                           (const short*)nullptr, stride, offsets, lines, oscale, smr, (double*)nullptr, bandPeak,
                           (const int*)nullptr, lay, sens);
    else
        hipLaunchKernelGGL((smr_kernel<false, double, 256, 1024, 1>), grid, dim3(256), lds, st, S, 1, (const double*)chL,
                           (const double*)nullptr, stride, offsets, lines, oscale, smr, (double*)nullptr, bandPeak,
                           (const int*)nullptr, lay, sens);
    return hipGetLastError();
}
#endif
#else   // the main unit

#ifdef MRC_NODE_STATS
extern "C" int mrc_debug_node_stats(unsigned long long* out4, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(gNodeStats), sizeof(unsigned long long) * 4);
    if (e == hipSuccess && reset) {
        unsigned long long z[4] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(gNodeStats), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

#ifdef MRC_PROFILE_PHASES
extern "C" int mrc_debug_phase_cycles(unsigned long long* out32, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(gPhaseCycles), sizeof(unsigned long long) * 32);
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(gPhaseCycles), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_smr(const DevShape& S, int64_t nFrames, const void* chL, const void* chR, int fmt, int64_t stride,
                      const int64_t* offsets, const double* lines, const int* oscale, double* smr, double* thresh,
                      double* bandPeak, const int* msSwitch, bool exactSpread, hipStream_t st, unsigned long long* sens) {
    if (nFrames <= 0) return hipSuccess;
    const int nsig = chR ? 4 : 1;
    const int H = S.H, M = S.halfN;
    size_t lds = 0;
    const SmrLds lay = smr_launch_layout(S, &lds);
    // blocks of up to 128 lines (two 64-line chunks) run as two-wave workgroups: no idle waves holding CU wave slots
    const dim3 grid((unsigned)(nFrames * nsig));
#define MRC_SMR_LAUNCH(EX, TY, THREADS, LG, MD)                                                                      \
    hipLaunchKernelGGL((smr_kernel<EX, TY, THREADS, LG, MD>), grid, dim3(THREADS), lds, st, S, nsig, (const TY*)chL, \
                       (const TY*)chR, stride, offsets, lines, oscale, smr, thresh, bandPeak, msSwitch, lay, sens)
    const bool isLong = H == 1024 && M == 1024 && S.peakLast == 924 && MRC_SMR_THREADS == 256 && lay.twOff >= 0;
    const bool isShort = H == 128 && M == 128 && S.peakLast == 28 && lay.twOff >= 0;
    const bool isTrans = H == 576 && M == 576 && S.peakLast == 476 && lay.twOff < 0 && MRC_SMR_THREADS == 256;
    // the hot paths: mono, and (long blocks) joint stereo with the switch known; no thresholds wanted
    const int mode = (thresh || !bandPeak) ? 0 : (nsig == 1 && !msSwitch) ? 1 : (nsig == 4 && msSwitch) ? 2 : 0;
#ifndef MRC_SMR_SHORT_LEAN                       // 1: short blocks of the hot paths run smr_short_kernel (a wavefront per unit)
#define MRC_SMR_SHORT_LEAN 1
#endif
    if (MRC_SMR_SHORT_LEAN && isShort && !exactSpread && mode != 0 && kExpTab == 64 && S.nBands <= 32 && S.N == 256 && !sens) {
        const int64_t nUnits = nFrames * nsig;
        const int run = (int)std::min<int64_t>(16, std::max<int64_t>(1, nUnits / (kShortWaves * 4096)));
        const unsigned g = (unsigned)((nUnits + (int64_t)kShortWaves * run - 1) / ((int64_t)kShortWaves * run));
        const size_t ldsS = (size_t)(kShortWaves * kShortWaveLds + kShortSharedLds) * sizeof(double);
#define MRC_SMR_SHORT(TY, MD)                                                                                          \
    hipLaunchKernelGGL((smr_short_kernel<TY, MD>), dim3(g), dim3(kWave * kShortWaves), ldsS, st, S, nUnits, run,        \
                       (const TY*)chL, (const TY*)chR, stride, offsets, lines, oscale, smr, bandPeak, msSwitch)
        if (fmt == kSampleI16) { if (mode == 1) MRC_SMR_SHORT(short, 1); else MRC_SMR_SHORT(short, 2); }
        else { if (mode == 1) MRC_SMR_SHORT(double, 1); else MRC_SMR_SHORT(double, 2); }
#undef MRC_SMR_SHORT
        return hipGetLastError();
    }
#if MRC_SMR_SPLIT                                // (mode 1: mono, no switch, no thresholds -- the other unit's kernel)
#define MRC_SMR_MONO_LONG(EX, TY) return launch_smr_mono_long(S, nFrames, chL, fmt, stride, offsets, lines, oscale, smr, bandPeak, st, sens)
#else
#define MRC_SMR_MONO_LONG(EX, TY) MRC_SMR_LAUNCH(EX, TY, MRC_SMR_THREADS, 1024, 1)
#endif
#define MRC_SMR_PICK(EX, TY) do { if (isShort && !EX && mode == 1) MRC_SMR_LAUNCH(EX, TY, 128, 128, 1);               \
                                  else if (isShort && !EX) MRC_SMR_LAUNCH(EX, TY, 128, 128, 0);                      \
                                  else if (M <= 2 * kWave) MRC_SMR_LAUNCH(EX, TY, 128, 0, 0);                        \
                                  else if (isLong && !EX && mode == 1) MRC_SMR_MONO_LONG(EX, TY);                    \
                                  else if (isLong && !EX && mode == 2) MRC_SMR_LAUNCH(EX, TY, MRC_SMR_THREADS, 1024, 2);   \
                                  else if (isLong && !EX) MRC_SMR_LAUNCH(EX, TY, MRC_SMR_THREADS, 1024, 0);          \
                                  else if (isTrans && !EX && mode == 1) MRC_SMR_LAUNCH(EX, TY, MRC_SMR_THREADS, 576, 1);   \
                                  else if (isTrans && !EX) MRC_SMR_LAUNCH(EX, TY, MRC_SMR_THREADS, 576, 0);          \
                                  else MRC_SMR_LAUNCH(EX, TY, MRC_SMR_THREADS, 0, 0); } while (0)
    if (fmt == kSampleI16) { if (exactSpread) MRC_SMR_PICK(true, short); else MRC_SMR_PICK(false, short); }
    else { if (exactSpread) MRC_SMR_PICK(true, double); else MRC_SMR_PICK(false, double); }
#undef MRC_SMR_PICK
#undef MRC_SMR_MONO_LONG
#undef MRC_SMR_LAUNCH
    return hipGetLastError();
}
#endif  // the main unit

}  // namespace mrc
