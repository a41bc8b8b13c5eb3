// CPU harness for mrcaudiocodec_amd/csrc/mrc_log10.hpp (the same source the device code compiles): compares
// log10_tab32 with long double log10l over the dynamic range the masking model uses and close to 1.
// Prints: max error in ulp for |log10 x| >= 1/4, max absolute error elsewhere.
#include "mrc_log10.hpp"
#include <cstdio>
#include <random>

int main() {
    double tab[mrc::kLogTabEntries * 4] = {};
    for (int j = 0; j < mrc::kLogTabEntries; ++j)
        for (int c = 0; c < 3; ++c) tab[4 * j + c] = mrc::kLog10Tab[j][c];
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> wide(-300, 300), near(-0.01, 0.01);
    double maxUlp = 0, maxAbs = 0;
    for (long i = 0; i < 4000000; ++i) {
        const double x = (i & 1) ? std::pow(10.0, wide(g)) : 1.0 + near(g) * ((i & 2) ? 1 : 0.001);
        const long double ref = log10l((long double)x);
        const double got = mrc::log10_tab32(x, tab);
        const double err = std::fabs((double)((long double)got - ref));
        const double mag = std::fabs((double)ref);
        if (mag >= 0.25) {
            int ex;
            std::frexp(mag, &ex);
            const double ulp = std::ldexp(1.0, ex - 53);
            if (err / ulp > maxUlp) maxUlp = err / ulp;
        } else if (err > maxAbs) {
            maxAbs = err;
        }
    }
    std::printf("%.4f %.4g\n", maxUlp, maxAbs);
    return 0;
}
