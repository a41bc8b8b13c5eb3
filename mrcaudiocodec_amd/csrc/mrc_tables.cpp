// Host-side constant tables per block shape (a,b): windows, twiddles, band tables, psychoacoustic
// constants, bit budgets.  Built once at mrc_create / first use of a shape and uploaded as one
// device blob.  Reference lines are cited per table; everything is a pure function of
// (a, b, sampleRate, nScaleBits, nMantSizeBits, targetBitsPerSample, blksw bits).
#include "mrc_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mrc {

namespace {

const long double kPiL = 3.14159265358979323846264338327950288L;

// Modified Bessel function I0 by its power series (all terms positive, no cancellation), long double.
long double bessel_i0(long double x) {
    long double q = x * x / 4.0L, term = 1.0L, sum = 1.0L;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((long double)k * (long double)k);
        sum += term;
        if (term < 1e-24L * sum) break;
    }
    return sum;
}

// window.py:49-101 -- Kaiser-Bessel-derived window, alpha = 4, length N: kernel
// w[j] = I0(pi a sqrt(1-((j-M/2)/(M/2))^2))/I0(pi a), j = 0..M = N/2; rising half
// sqrt(sum_{j<=n} w^2 / sum_{0..M} w^2), falling half sqrt(sum_{j>=i+1} w^2 / total).
std::vector<double> kbd_table(int N, long double alpha = 4.0L) {
    const int M = N / 2;
    std::vector<long double> w2(M + 1);
    const long double den = bessel_i0(kPiL * alpha);
    long double total = 0.0L;
    for (int j = 0; j <= M; ++j) {
        long double r = ((long double)j - M / 2.0L) / (M / 2.0L);
        long double rad = 1.0L - r * r;
        if (rad < 0.0L) rad = 0.0L;
        long double k = bessel_i0(kPiL * alpha * sqrtl(rad)) / den;
        w2[j] = k * k;
        total += w2[j];
    }
    std::vector<double> win(N);
    long double run = 0.0L;
    for (int n = 0; n < M; ++n) {
        run += w2[n];
        win[n] = (double)sqrtl(run / total);
    }
    run = 0.0L;
    for (int i = M - 1; i >= 0; --i) {
        run += w2[i + 1];
        win[M + i] = (double)sqrtl(run / total);
    }
    return win;
}

double thresh_quiet_db(double f) {                    // psychoac.py:20-25
    double k = f / 1000.;
    double d = k - 3.3;
    return (3.64 * std::pow(k, -0.8)) - (6.5 * std::exp(-0.6 * (d * d))) + (0.001 * std::pow(k, 4.0));
}

double bark(double f) {                               // psychoac.py:27-29
    double q = f / 7500.;
    return 13 * std::atan(0.76 * f / 1000.) + 3.5 * std::atan(q * q);
}

bool factor(int n, int* rad, int* nrad) {
    int c = 0;
    while (n % 4 == 0 && c < kMaxRadices) { rad[c++] = 4; n /= 4; }
    while (n % 2 == 0 && c < kMaxRadices) { rad[c++] = 2; n /= 2; }
    while (n % 3 == 0 && c < kMaxRadices) { rad[c++] = 3; n /= 3; }
    *nrad = c;
    return n == 1;
}

// psychoac.py:82-84 and pacfileThem.py:643
const int kLongLimits[25] = {100, 200, 300, 400, 510, 630, 770, 920, 1080, 1270, 1480, 1720, 2000, 2320,
                             2700, 3150, 3700, 4400, 5300, 6400, 7700, 9500, 12000, 15500, 24000};
const int kShortLimits[9] = {300, 630, 1080, 1720, 2700, 4400, 7700, 15500, 24000};

struct BlobWriter {
    std::vector<unsigned char> bytes;
    template <class T> size_t put(const std::vector<T>& v) {
        size_t off = (bytes.size() + 15) & ~size_t(15);
        bytes.resize(off + v.size() * sizeof(T));
        std::memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
};

std::vector<double2> unit_circle(int n, long double scale /* angle = -scale * t */) {
    std::vector<double2> w(n);
    for (int t = 0; t < n; ++t) {
        long double ang = -scale * (long double)t;
        w[t] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    return w;
}

}  // namespace

// quantize.py:114-146 on the host (used for validation of configs only; the device has its own copy).
int scale_factor_host(double v, int nScaleBits, int nMantBits) {
    int nBits = (1 << nScaleBits) - 1 + nMantBits;
    double mag = std::fabs(v);
    long long code;
    if (mag >= 1.0) code = (1LL << (nBits - 1)) - 1;
    else code = (long long)((((double)((1LL << nBits) - 1)) * mag + 1.0) / 2.0);
    int top = 0;
    if (code > 0) top = 63 - __builtin_clzll((unsigned long long)code);
    int lz = (nBits - 2) - top;
    int cap = (1 << nScaleBits) - 1;
    return lz < cap ? lz : cap;
}

// psychoac.py:86-105 with the limits chosen at pacfileThem.py:637-645: 25 critical bands for long+long
// blocks, the 9-band table for every other shape.  Band i takes the not-yet-assigned lines whose centre
// (n+1/2) fs/(2 halfN) lies below limit i; the last band takes the rest.
bool band_table(const mrc_config& cfg, int a, int b, std::vector<int>* count) {
    const int N = a + b, halfN = N / 2;
    const bool isLong = (N == 2 * cfg.n_mdct_lines);
    const int* lim = isLong ? kLongLimits : kShortLimits;
    const int nb = isLong ? 25 : 9;
    count->assign(nb, 0);
    int j = 0;
    for (int i = 0; i < nb - 1; ++i)
        while (j < halfN && (j + 0.5) * (((double)cfg.sample_rate / halfN) / 2.) < lim[i]) { ++(*count)[i]; ++j; }
    (*count)[nb - 1] = halfN - j;
    return (*count)[nb - 1] >= 0;
}

void ms_plan(const std::vector<int>& bandLo, const std::vector<int>& bandN, std::vector<int>* plan, int* nLeaves,
             int* nInternal) {
    std::vector<std::pair<int, int>> leaves, inner;        // (lo, n); (left node, right node)
    struct Rec {
        std::vector<std::pair<int, int>>& leaves;
        std::vector<std::pair<int, int>> pending;           // internal nodes as (left id, right id), ids fixed up below
        // returns a node reference: >= 0 leaf index, < 0 -(internal index + 1)
        int build(int lo, int n) {
            if (n <= 128) { leaves.push_back({lo, n}); return (int)leaves.size() - 1; }
            int n2 = n / 2;
            n2 -= n2 % 8;
            const int l = build(lo, n2);
            const int r = build(lo + n2, n - n2);
            pending.push_back({l, r});
            return -(int)pending.size();
        }
    } rec{leaves, {}};
    std::vector<int> roots;
    for (size_t b = 0; b < bandN.size(); ++b) roots.push_back(rec.build(bandLo[b], bandN[b]));
    const int nl = (int)leaves.size();
    // the kernels take the leaves eight at a time, a round lasting as long as its longest leaf: longest first, so that the
    // rounds are as even as they can be (the order of the LEAVES is free: every sum keeps its own order of additions)
    std::vector<int> order(nl), newId(nl);
    for (int i = 0; i < nl; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return leaves[x].second > leaves[y].second; });
    for (int i = 0; i < nl; ++i) newId[order[i]] = i;
    auto id = [nl, &newId](int ref) { return ref >= 0 ? newId[ref] : nl + (-ref - 1); };
    plan->clear();
    for (int i = 0; i < nl; ++i) { plan->push_back(leaves[order[i]].first); plan->push_back(leaves[order[i]].second); }
    for (auto& in : rec.pending) { plan->push_back(id(in.first)); plan->push_back(id(in.second)); }
    for (int r : roots) plan->push_back(id(r));
    *nLeaves = nl;
    *nInternal = (int)rec.pending.size();
}

bool build_shape(const mrc_config& cfg, int a, int b, HostShape* out, std::string* err) {
    const int N = a + b;
    if (a <= 0 || b <= 0 || N % 4 != 0 || (b - a) % 4 != 0) {
        *err = "block shape (a,b) must be positive with a+b and b-a divisible by 4";
        return false;
    }
    DevShape& S = out->dev;
    S = DevShape{};
    S.a = a; S.b = b; S.N = N; S.halfN = N / 2; S.Q = N / 4; S.H = N / 2;
    S.shift = (b - a) / 4;
    S.peakLast = N / 2 - 100;
    if (S.peakLast < 3) { *err = "block too short for the peak search (N/2-100 < 3)"; return false; }
    if (!factor(S.Q, S.radQ, &S.nRadQ) || !factor(S.H, S.radH, &S.nRadH)) {
        *err = "block length must factor into 2s and 3s";
        return false;
    }
    S.nScaleBits = cfg.n_scale_bits;
    S.maxMantBits = (1 << cfg.n_mant_size_bits) > 16 ? 16 : (1 << cfg.n_mant_size_bits);   // codecThem.py:292-293
    S.twoOverN = 2.0 / N;
    S.binHz = (double)(cfg.sample_rate / N);                  // py2 integer division (psychoac.py:165)
    S.xiDen = ((double)N * (double)N) * (3. / 8.);

    // band table: psychoac.py:86-105 with the limits chosen at pacfileThem.py:637-645
    std::vector<int> count;
    if (!band_table(cfg, a, b, &count)) { *err = "band table overflow"; return false; }
    const int nb = (int)count.size();
    S.nBands = nb;
    out->bandN = count;
    out->bandLo.assign(nb, 0);
    for (int i = 1; i < nb; ++i) out->bandLo[i] = out->bandLo[i - 1] + count[i - 1];
    std::vector<unsigned char> bandOfLine(S.halfN);
    for (int i = 0; i < nb; ++i)
        for (int k = 0; k < count[i]; ++k) bandOfLine[out->bandLo[i] + k] = (unsigned char)i;

    // bit budgets: codecThem.py:299-306 (mono) and 381-388 (joint, before the reservoir is added)
    {
        double halfN = (a + b) / 2.;
        double m = cfg.target_bits_per_sample * halfN;
        m -= cfg.n_scale_bits * (nb + 1);
        m -= cfg.n_mant_size_bits * nb;
        m -= cfg.blksw_bits_a;
        m -= cfg.blksw_bits_b;
        S.budgetMono = m;
        double j = cfg.target_bits_per_sample * halfN;
        j -= cfg.n_scale_bits * nb;
        j -= cfg.n_mant_size_bits * nb;
        j += j;
        j -= nb;
        j -= cfg.n_scale_bits * 4;
        S.budgetJointPre = j;
        S.blkswA = cfg.blksw_bits_a;
        S.blkswB = cfg.blksw_bits_b;
    }

    // windows
    std::vector<double> win(N), hann(N);
    {
        std::vector<double> ka = kbd_table(2 * a), kb = (a == b) ? ka : kbd_table(2 * b);
        for (int n = 0; n < a; ++n) win[n] = ka[n];
        for (int i = 0; i < b; ++i) win[a + i] = kb[b + i];
        for (int n = 0; n < N; ++n)                      // window.py:38-42
            hann[n] = 0.5 + (-0.5 * std::cos(((2.0 * M_PI) / N) * (n + 0.5)));
        // a = b: the two halves are built from the same running sums in the same order, so win[n] == win[N - 1 - n] bit for
        // bit; checked rather than assumed (mdct_long_kernel keeps only one half of the window in registers when it holds)
        S.winSymmetric = 1;
        for (int n = 0; n < N / 2; ++n)
            if (std::memcmp(&win[n], &win[N - 1 - n], sizeof(double)) != 0) S.winSymmetric = 0;
    }
    // twiddles
    const int M = S.halfN;
    std::vector<double2> pre(S.Q), post(S.Q);
    for (int n = 0; n < S.Q; ++n) {
        long double ang = -kPiL * (4.0L * n + 1.0L) / (4.0L * M);
        pre[n] = make_double2((double)cosl(ang), (double)sinl(ang));
        long double ang2 = -kPiL * (long double)n / (long double)M;
        post[n] = make_double2((double)cosl(ang2), (double)sinl(ang2));
    }
    std::vector<double2> wQ = unit_circle(S.Q, 2.0L * kPiL / S.Q);
    std::vector<double2> wH = unit_circle(S.H, 2.0L * kPiL / S.H);
    std::vector<double2> wN = unit_circle(S.H, 2.0L * kPiL / N);
    // psychoacoustic constants on the MDCT line grid (psychoac.py:142-143,155)
    std::vector<double> zb(S.halfN), quiet(S.halfN), lowE(S.halfN);
    const long double lowBits = 2.7L * log2l(10.0L);     // 27 dB/Bark (psychoac.py:74) in bits per Bark
    for (int k = 0; k < S.halfN; ++k) {
        double f = (k + 0.5) * (((double)cfg.sample_rate / S.halfN) / 2.);
        zb[k] = bark(f);
        quiet[k] = std::pow(10.0, (thresh_quiet_db(f) - 96) / 10);
        lowE[k] = (double)powl(2.0L, lowBits * ((long double)zb[k] + 0.5L));
    }

    // search hints for the masker-side line searches of smr_kernel: for a masker near line k the first line that
    // sees it / the first line more than 1/2 Bark above it are close to these (the kernel fixes them up exactly)
    std::vector<unsigned short> loLine(S.halfN), hiLine(S.halfN);
    for (int k = 0, lo = 0, hi = 0; k < S.halfN; ++k) {
        while (lo < S.halfN && !(zb[lo] - zb[k] >= -0.5)) ++lo;
        while (hi < S.halfN && !(zb[hi] - zb[k] > 0.5)) ++hi;
        loLine[k] = (unsigned short)lo;
        hiLine[k] = (unsigned short)hi;
    }
    S.linesPerHz = (double)N / (double)cfg.sample_rate;       // line index ~ f * N/fs - 1/2

    std::vector<int> msPlanV;
    ms_plan(out->bandLo, out->bandN, &msPlanV, &S.msLeaves, &S.msInternal);
    if (S.msLeaves + S.msInternal > 64) { *err = "band table too fine for the M/S summation plan"; return false; }

    // per-pass twiddles of the 1024-point psycho FFT: for the passes p = 4, 16, 64, 256 and k < p the factors
    // w^(k r 1024 / (4 p)), r = 1..3, taken from the FIRST QUADRANT of wH with the fix-ups w(s + q n/4) = w(s) (-i)^q -- the
    // values dev::TwQuarter hands out (fft_lds_1024), so that both transforms agree bit for bit
    std::vector<double2> fftTw;
    if (S.H == 1024) {
        for (int p = 4; p <= 256; p *= 4)
            for (int k = 0; k < p; ++k)
                for (int r = 1; r <= 3; ++r) {
                    const int t = k * r * (1024 / (4 * p));
                    const double2 v = wH[t & 255];
                    const int q = t >> 8;
                    double2 o = (q & 1) ? make_double2(v.y, -v.x) : v;
                    if (q & 2) o = make_double2(-o.x, -o.y);
                    fftTw.push_back(o);
                }
    }
    std::vector<LineConstants> lineC(S.halfN);
    for (int k = 0; k < S.halfN; ++k) lineC[k] = LineConstants{zb[k], quiet[k], lowE[k], (int)bandOfLine[k], 0};

    BlobWriter bw;
    size_t oWin = bw.put(win), oHann = bw.put(hann), oPre = bw.put(pre), oPost = bw.put(post);
    size_t oWQ = bw.put(wQ), oWH = bw.put(wH), oWN = bw.put(wN), oZb = bw.put(zb), oQuiet = bw.put(quiet), oLowE = bw.put(lowE);
    size_t oLo = bw.put(out->bandLo), oCnt = bw.put(out->bandN), oBol = bw.put(bandOfLine);
    size_t oLoLine = bw.put(loLine), oHiLine = bw.put(hiLine), oMsPlan = bw.put(msPlanV);
    size_t oLineC = bw.put(lineC), oFftTw = bw.put(fftTw);
    void* blob = nullptr;
    if (hipMalloc(&blob, bw.bytes.size()) != hipSuccess) { *err = "hipMalloc(shape tables) failed"; return false; }
    if (hipMemcpy(blob, bw.bytes.data(), bw.bytes.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(blob);
        *err = "hipMemcpy(shape tables) failed";
        return false;
    }
    unsigned char* base = (unsigned char*)blob;
    S.win = (const double*)(base + oWin);       S.hann = (const double*)(base + oHann);
    S.pre = (const double2*)(base + oPre);      S.post = (const double2*)(base + oPost);
    S.wQ = (const double2*)(base + oWQ);        S.wH = (const double2*)(base + oWH);
    S.wN = (const double2*)(base + oWN);
    S.zb = (const double*)(base + oZb);         S.quiet = (const double*)(base + oQuiet);
    S.lowE = (const double*)(base + oLowE);
    S.bandLo = (const int*)(base + oLo);        S.bandN = (const int*)(base + oCnt);
    S.bandOfLine = (const unsigned char*)(base + oBol);
    S.loLine = (const unsigned short*)(base + oLoLine);
    S.hiLine = (const unsigned short*)(base + oHiLine);
    S.msPlan = (const int*)(base + oMsPlan);
    S.lineC = (const LineConstants*)(base + oLineC);
    S.fftTw = fftTw.empty() ? nullptr : (const double2*)(base + oFftTw);
    out->blob = blob;
    return true;
}

void free_shape(HostShape* s) {
    if (s->blob) (void)hipFree(s->blob);
    s->blob = nullptr;
}

}  // namespace mrc
