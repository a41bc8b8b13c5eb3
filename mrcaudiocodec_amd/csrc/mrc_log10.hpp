// log10 for the SPL conversions of the masking model (psychoac.py:8-12: 96 + 10 log10(intensity)), three per
// line and masker.  The runtime library's log10 costs ~90 fp64 instructions; this one ~30:
//     x = 2^e m, m in [1/2, 1);  j = top five mantissa bits;  r = fma(m, 1/c_j, -1), |r| <= 1/64 (exact up to
//     one rounding of 2^-59);  log10 x = e log10 2 + log10 c_j + log1p(r)/ln 10,  Taylor to r^9.
// log10 c_j and log10 2 are carried as hi + lo, the leading product r/ln10 with its rounding error, so the
// result is within 0.6 ulp of the exact value for |log10 x| >= 1/4 and within 4e-17 ABSOLUTE near x = 1
// (what matters here: the value is added to 96).  Compiles for the host as well (tests/test_abi.py checks it
// against long double log10l on a CPU build).  Only normal positive x: callers fall back otherwise.
#pragma once
#include <cmath>
#include <cstring>

#if defined(__HIPCC__)
#define MRC_HD __host__ __device__ __forceinline__
#else
#define MRC_HD inline
#endif

namespace mrc {

constexpr int kLogTabEntries = 32;
// {1/c_j rounded, log10(c_j) hi, lo} with c_j := 1 / (1/c_j rounded), c_j ~ 1/2 + (j + 1/2)/64
constexpr double kLog10Tab[kLogTabEntries][3] = {
    {0x1.f81f81f81f820p+0, -0x1.2d5c1760b86bbp-2, -0x1.36eb80b65a8bep-57},
    {0x1.e9131abf0b767p+0, -0x1.1fe1e5af2c141p-2, 0x1.d4e3b1be4a899p-57},
    {0x1.dae6076b981dbp+0, -0x1.12cd31b9c99ffp-2, -0x1.79e1f24993339p-60},
    {0x1.cd85689039b0bp+0, -0x1.06182e84fd4acp-2, -0x1.9e35c3622c873p-60},
    {0x1.c0e070381c0e0p+0, -0x1.f37b15bab08d0p-3, -0x1.fc1e600afd667p-59},
    {0x1.b4e81b4e81b4fp+0, -0x1.db70c7e96e7f4p-3, -0x1.b5c84186a7553p-61},
    {0x1.a98ef606a63bep+0, -0x1.c4087384f4f81p-3, -0x1.f9667c0d44470p-58},
    {0x1.9ec8e951033d9p+0, -0x1.ad39c9c2c607fp-3, -0x1.748e5120bfdeep-57},
    {0x1.948b0fcd6e9e0p+0, -0x1.96fd1b639fc08p-3, -0x1.3439ecb3a6f4ap-58},
    {0x1.8acb90f6bf3aap+0, -0x1.814b4921bd52cp-3, -0x1.e2b0ac1a89094p-58},
    {0x1.8181818181818p+0, -0x1.6c1db5f9bb335p-3, -0x1.b0734786a535ep-57},
    {0x1.78a4c8178a4c8p+0, -0x1.576e3b0bde0a7p-3, 0x1.af3625c6d1951p-58},
    {0x1.702e05c0b8170p+0, -0x1.43371cde076c1p-3, -0x1.65f2740783efep-57},
    {0x1.6816816816817p+0, -0x1.2f7301cf4e87cp-3, -0x1.c37b5df960585p-59},
    {0x1.6058160581606p+0, -0x1.1c1ce9955c0c7p-3, -0x1.0c6f7d84b8adbp-61},
    {0x1.58ed2308158edp+0, -0x1.093025a19976bp-3, -0x1.eb0147c0a50ccp-58},
    {0x1.51d07eae2f815p+0, -0x1.ed50a4a26eafbp-4, -0x1.7b3241e2c090cp-58},
    {0x1.4afd6a052bf5bp+0, -0x1.c902a19e65114p-4, -0x1.c89d9d30df676p-60},
    {0x1.446f86562d9fbp+0, -0x1.a56e8325f5c87p-4, -0x1.40d233375080ep-58},
    {0x1.3e22cbce4a902p+0, -0x1.828cfed29a212p-4, 0x1.a821d60e7beb0p-62},
    {0x1.3813813813814p+0, -0x1.605735ee985f4p-4, 0x1.cc64bbd528d96p-59},
    {0x1.323e34a2b10bfp+0, -0x1.3ec6ad5407866p-4, -0x1.b9afeaf9d54b2p-60},
    {0x1.2c9fb4d812ca0p+0, -0x1.1dd5460c8b170p-4, -0x1.f81ad1789f5f8p-58},
    {0x1.27350b8812735p+0, -0x1.fafa6d397efdbp-5, 0x1.d327d1330f1b3p-59},
    {0x1.21fb78121fb78p+0, -0x1.bb7209d1e24e4p-5, -0x1.d186075dd2453p-59},
    {0x1.1cf06ada2811dp+0, -0x1.7d070145f4fd8p-5, -0x1.bc7c3803efe02p-62},
    {0x1.1811811811812p+0, -0x1.3faf7c6630614p-5, 0x1.b29bcb7e05666p-61},
    {0x1.135c81135c811p+0, -0x1.0362241e638eap-5, 0x1.5dc738afb05e2p-59},
    {0x1.0ecf56be69c90p+0, -0x1.902c31d62a847p-6, 0x1.124bcf1db4169p-62},
    {0x1.0a6810a6810a7p+0, -0x1.1b85d6044e9bbp-6, -0x1.eca1cabde3904p-61},
    {0x1.0624dd2f1a9fcp+0, -0x1.51824c7587eb5p-7, -0x1.92913f4597c39p-64},
    {0x1.0204081020408p+0, -0x1.be76bd77b4fb5p-9, -0x1.3d795a9dbba76p-64}};

// tab: kLog10Tab laid out [j][4] = {1/c, log10 c hi, lo, unused} (the device copy sits in LDS)
MRC_HD double log10_tab32(double x, const double* tab) {
    constexpr double kL2Hi = 0x1.34413509f7800p-2;       // log10(2): 42 significant bits, e * kL2Hi is exact
    constexpr double kL2Lo = 0x1.fef311f12b358p-46;
    constexpr double kC1Hi = 0x1.bcb7b1526e50ep-2;       // 1/ln(10) = hi + lo
    constexpr double kC1Lo = 0x1.95355baaafad3p-57;
    int e;
    const double m = frexp(x, &e);
    long long bits;
    memcpy(&bits, &m, sizeof bits);
    const int j = (int)(bits >> 47) & 31;
    const double invc = tab[4 * j], lh = tab[4 * j + 1], ll = tab[4 * j + 2];
    const double r = fma(m, invc, -1.0);
    const double ed = (double)e;
    // log1p(r)/ln10 = r/ln10 + r^2 (c2 + c3 r + ... + c9 r^7)
    double p = 0x1.8b4df2f3f047ep-5;
    p = fma(p, r, -0x1.bcb7b1526e50ep-5);
    p = fma(p, r, 0x1.fc3fa615105c7p-5);
    p = fma(p, r, -0x1.287a7636f435fp-4);
    p = fma(p, r, 0x1.63c62775250d8p-4);
    p = fma(p, r, -0x1.bcb7b1526e50ep-4);
    p = fma(p, r, 0x1.287a7636f435fp-3);
    p = fma(p, r, -0x1.bcb7b1526e50ep-3);
    const double s = ed * kL2Hi;                         // exact
    const double hi = s + lh;                            // |s| >= |lh| or s == 0: fast two-sum
    const double err = (s - hi) + lh;
    const double t = r * kC1Hi;
    const double tl = fma(r, kC1Hi, -t);
    double lo = fma(r, kC1Lo, tl);
    lo = fma(ed, kL2Lo, lo + (ll + err));
    lo = fma(r * r, p, lo);
    return hi + (t + lo);
}

}  // namespace mrc
