// Host-side back end of the encoder: Huffman table choice + recoding and `.pac` bit packing.
// BASELINE.json's north_star keeps huffman.py / bitpack.py on the host; this is their C++ form so that the
// GPU path is not throttled by per-bit Python (bitpack.py:36-101) and per-block pickle loads
// (codecThem.py:151-153).  No GPU, no handle: pure functions of the dense outputs of mrc_encode_*.
//
//   Huffman gain / table choice   codecThem.py:136-203
//   chunk sizes                   pacfileThem.py:652-707 (independent channels), 825-880 (joint)
//   chunk payloads                pacfileThem.py:716-781, 892-963; bit order of bitpack.py:36-101
//   file header                   pacfileThem.py:586-613
// Table ids: sorted table names (percussive 0, silence 1, speech 2, tonal 3), 15 = raw; see DESIGN.md.
#include "mrc_internal.hpp"

#include <sched.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

namespace {

struct HuffTable {
    int escape;
    int nCodes;
    struct { int value; const char* code; } codes[16];
};

// training_data/*_table.pkl of the reference, as data (tools/check_huffman_tables.py)
const HuffTable kTables[4] = {
    {16, 16, {{0, "0"}, {1, "1110"}, {2, "110"}, {3, "111101"}, {4, "101"}, {5, "1000"}, {6, "111100"},
              {7, "11111100"}, {8, "10010"}, {9, "111110"}, {10, "1111111"}, {11, "1001101"}, {12, "1001100"},
              {13, "111111011"}, {14, "111111010"}, {16, "100111"}}},
    {11, 11, {{0, "11"}, {1, "000"}, {2, "100"}, {3, "00101"}, {4, "01"}, {5, "0011"}, {6, "101101"},
              {8, "10111"}, {9, "00100"}, {10, "101100"}, {11, "1010"}}},
    {7, 16, {{0, "11"}, {1, "1001"}, {2, "101"}, {3, "100011"}, {4, "00"}, {5, "0100"}, {6, "100000"},
             {7, "0101"}, {8, "0111"}, {9, "01101"}, {10, "100001"}, {11, "1000101"}, {12, "0110000"},
             {16, "011001"}, {17, "1000100"}, {32, "0110001"}}},
    {7, 16, {{0, "0"}, {1, "11110"}, {2, "110"}, {3, "1111101"}, {4, "101"}, {5, "1001011"}, {6, "11111101"},
             {7, "1000"}, {8, "1110"}, {9, "1111111"}, {10, "10010100"}, {16, "10011"}, {17, "1111100"},
             {18, "11111100"}, {32, "100100"}, {64, "10010101"}}},
};
constexpr int kRawTable = 15;       // codecThem.py:149
constexpr int kLutSize = 65;        // largest table value is 64

struct Lut {
    unsigned char len[4][kLutSize];
    unsigned short bits[4][kLutSize];
    // Pricing all four tables in ONE pass over the mantissas: per value two 64-bit words of four 16-bit fields
    // (field t = table t): lenSum[v] = code length, 0 where the table has no code for v; miss[v] = 1 there.
    // A band's sums stay below 2^16 (<= 1024 lines x 9 bits).  Index kLutSize = "any larger value": in no table.
    uint64_t lenSum[kLutSize + 1], miss[kLutSize + 1];
    // Emission: emit[t][v] = code | length << 16 | (1 << 31 where the raw mantissa follows: escape value / no code)
    uint32_t emit[4][kLutSize + 1];
    Lut() {
        std::memset(len, 0, sizeof(len));
        std::memset(bits, 0, sizeof(bits));
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < kTables[t].nCodes; ++i) {
                const char* c = kTables[t].codes[i].code;
                unsigned v = 0;
                int n = 0;
                for (; c[n]; ++n) v = (v << 1) | (unsigned)(c[n] - '0');
                len[t][kTables[t].codes[i].value] = (unsigned char)n;
                bits[t][kTables[t].codes[i].value] = (unsigned short)v;
            }
        for (int v = 0; v <= kLutSize; ++v) {
            lenSum[v] = miss[v] = 0;
            for (int t = 0; t < 4; ++t) {
                const int esc = kTables[t].escape;
                const int n = v < kLutSize ? len[t][v] : 0;
                lenSum[v] |= (uint64_t)n << (16 * t);
                miss[v] |= (uint64_t)(n == 0) << (16 * t);
                if (n != 0 && v != esc) emit[t][v] = bits[t][v] | ((uint32_t)n << 16);
                else emit[t][v] = bits[t][esc] | ((uint32_t)len[t][esc] << 16) | 0x80000000u;      // codecThem.py:194-200
            }
        }
    }
};
const Lut kLut;
inline unsigned lut_index(int32_t v) { return (uint32_t)v < (uint32_t)kLutSize ? (unsigned)v : (unsigned)kLutSize; }

// MSB-first writer, same byte image as bitpack.py:36-101 (zero-initialised buffer, bits OR-ed in)
struct BitWriter {
    uint8_t* p;
    uint64_t acc = 0;
    int n = 0;                                        // valid low bits of acc, < 32 between calls
    explicit BitWriter(uint8_t* dst) : p(dst) {}
    void put(uint32_t info, int nBits) {              // lowest nBits (<= 32) of info
        if (nBits <= 0) return;
        const uint64_t v = nBits >= 32 ? info : (info & ((1u << nBits) - 1u));
        acc = (acc << nBits) | v;
        n += nBits;
        if (n >= 32) {                                // four whole bytes, most significant first
            const uint32_t w = (uint32_t)(acc >> (n - 32));
            p[0] = (uint8_t)(w >> 24); p[1] = (uint8_t)(w >> 16); p[2] = (uint8_t)(w >> 8); p[3] = (uint8_t)w;
            p += 4;
            n -= 32;
        }
    }
    void flush() {
        while (n >= 8) { *p++ = (uint8_t)(acc >> (n - 8)); n -= 8; }
        if (n > 0) { *p++ = (uint8_t)(acc << (8 - n)); n = 0; }
    }
};

struct ChannelPlan {
    int table;          // 0..3 or 15
    int64_t mantBits;   // bits of the mantissa part as the writer emits them
    int bitsSaved;      // codecThem.py:202 (raw - best priced cost)
};

// codecThem.py:136-203.  The price of a table counts the escape VALUE itself as its code only
// (lines 169-172) although the writer emits code + raw mantissa for it (194-200): kept, so the choice and
// bits_saved equal the reference's; mantBits is what is really written (pacfileThem.py:685-703).
template <class MantT>
int64_t written_bits(const int32_t* ba, const MantT* mant, const std::vector<int>& nLines, int t) {
    // what the writer emits for table t (pacfileThem.py:685-703): code, plus the raw mantissa after an escape code
    const int nb = (int)nLines.size();
    int64_t written = 0;
    int k = 0;
    for (int b = 0; b < nb; ++b) {
        const int n = nLines[b];
        if (ba[b]) {
            int64_t codeBits = 0, rawFollows = 0;
            for (int i = 0; i < n; ++i) {
                const uint32_t e = kLut.emit[t][lut_index((int32_t)mant[k + i])];
                codeBits += (e >> 16) & 0x7fffu;
                rawFollows += e >> 31;
            }
            written += codeBits + rawFollows * ba[b];
        }
        k += n;
    }
    return written;
}

template <class MantT>
ChannelPlan plan_channel(const int32_t* ba, const MantT* mant, const std::vector<int>& nLines, int useHuffman,
                         int givenTable = -1) {
    const int nb = (int)nLines.size();
    int64_t raw = 0;
    for (int b = 0; b < nb; ++b)
        if (ba[b]) raw += (int64_t)ba[b] * nLines[b];
    ChannelPlan plan{kRawTable, raw, 0};
    if (givenTable >= 0) {                           // the table was chosen elsewhere (huffman_gain_kernel): no pricing
        if (givenTable != kRawTable) { plan.table = givenTable; plan.mantBits = written_bits(ba, mant, nLines, givenTable); }
        return plan;
    }
    if (!useHuffman) return plan;
    int64_t priced[4] = {0, 0, 0, 0};
    int k = 0;
    for (int b = 0; b < nb; ++b) {
        const int n = nLines[b];
        if (ba[b]) {
            uint64_t lens = 0, misses = 0;           // four 16-bit sums each
            for (int i = 0; i < n; ++i) {
                const unsigned idx = lut_index((int32_t)mant[k + i]);
                lens += kLut.lenSum[idx];
                misses += kLut.miss[idx];
            }
            for (int t = 0; t < 4; ++t)
                priced[t] += (int64_t)((lens >> (16 * t)) & 0xffffu) +
                             (int64_t)((misses >> (16 * t)) & 0xffffu) * (ba[b] + kLut.len[t][kTables[t].escape]);
        }
        k += n;
    }
    int64_t best = raw;
    for (int t = 0; t < 4; ++t)
        if (priced[t] < best) { best = priced[t]; plan.table = t; }
    if (plan.table != kRawTable) plan.mantBits = written_bits(ba, mant, nLines, plan.table);
    plan.bitsSaved = (int)(raw - best);
    return plan;
}

template <class MantT>
void write_band_records(BitWriter& w, const mrc_config& cfg, const int32_t* sf, const int32_t* ba, const MantT* mant,
                        const std::vector<int>& nLines, int table) {
    const int nb = (int)nLines.size();
    int k = 0;
    for (int b = 0; b < nb; ++b) {
        const int bits = ba[b];
        // pacfileThem.py:730-732: bit allocation (stored one lower) and scale factor, one put
        w.put(((uint32_t)(bits ? bits - 1 : 0) << cfg.n_scale_bits) | ((uint32_t)sf[b] & ((1u << cfg.n_scale_bits) - 1u)),
              cfg.n_mant_size_bits + cfg.n_scale_bits);
        const int n = nLines[b];
        if (bits) {
            const MantT* m = mant + k;
            if (table == kRawTable) {
                for (int i = 0; i < n; ++i) w.put((uint32_t)m[i], bits);
            } else {
                const uint32_t* emit = kLut.emit[table];
                const uint32_t rawMask = (1u << bits) - 1u;            // bits <= 16
                for (int i = 0; i < n; ++i) {
                    const uint32_t e = emit[lut_index((int32_t)m[i])];
                    const int len = (int)((e >> 16) & 0x7fffu);
                    if (e >> 31) w.put(((e & 0xffffu) << bits) | ((uint32_t)m[i] & rawMask), len + bits);   // code + raw: <= 9 + 16 bits
                    else w.put(e & 0xffffu, len);
                }
            }
        }
        k += n;
    }
}

// Worker threads of the packer / parser: the CPUs this process may run on, at most 16 (one GPU's share of a
// host), MRC_PACK_THREADS or mrc_pack_set_threads() override.
int default_pack_threads() {
    if (const char* e = std::getenv("MRC_PACK_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 1024) return v;
    }
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n > 16 ? 16 : n;
}
std::atomic<int> g_packThreads{default_pack_threads()};

// contiguous ranges over up to g_packThreads host threads (1 = run inline)
template <class F> void parallel_for(int64_t n, F body) {
    int nt = g_packThreads.load();
    // a thread is worth starting for ~16 chunks or more (a chunk packs in ~3 us, a thread starts and joins in ~25): one
    // block's two chunks on two fresh threads took 60 us instead of 7
    if (nt > n / 16) nt = (int)(n / 16);
    if (nt <= 1) { for (int64_t i = 0; i < n; ++i) body(i); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        pool.emplace_back([lo, hi, &body] { for (int64_t i = lo; i < hi; ++i) body(i); });
    }
    for (auto& th : pool) th.join();
}

inline void put_u32le(uint8_t* p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = v >> 24; }

bool shape_ok(const mrc_config* cfg, int a, int b) {
    return cfg && a > 0 && b > 0 && (a + b) % 2 == 0 && cfg->n_mdct_lines > 0 && cfg->n_scale_bits >= 1 &&
           cfg->n_scale_bits <= 4 && cfg->n_mant_size_bits >= 1 && cfg->n_mant_size_bits <= 8;
}


// ---- decode side ("next" row f-4): chunk parser -----------------------------------------------------------
// Huffman codes are at most 9 bits long: a 512-entry table per code maps the next 9 bits to (value, length).
constexpr int kPeekBits = 9;
struct DecLut {
    short value[4][1 << kPeekBits];
    unsigned char len[4][1 << kPeekBits];
    DecLut() {
        std::memset(len, 0, sizeof(len));
        std::memset(value, 0, sizeof(value));
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < kTables[t].nCodes; ++i) {
                const int v = kTables[t].codes[i].value, n = kLut.len[t][v];
                const unsigned head = (unsigned)kLut.bits[t][v] << (kPeekBits - n);
                for (unsigned tail = 0; tail < (1u << (kPeekBits - n)); ++tail) {
                    value[t][head | tail] = (short)v;
                    len[t][head | tail] = (unsigned char)n;
                }
            }
    }
};
const DecLut kDecLut;

// MSB-first reader (bitpack.py:104-170); reading past the end flags an error and returns zeros
struct BitReader {
    const uint8_t* p;
    int64_t nBits, pos = 0;
    bool ok = true;
    BitReader(const uint8_t* data, int64_t nBytes) : p(data), nBits(nBytes * 8) {}
    unsigned peek(int n) const {                       // n <= 25; bits past the end read as zero
        uint64_t w = 0;
        const int64_t byte = pos >> 3;
        for (int i = 0; i < 5; ++i) w = (w << 8) | ((byte + i) * 8 < nBits ? p[byte + i] : 0u);
        return (unsigned)((w >> (40 - (pos & 7) - n)) & ((1u << n) - 1u));
    }
    unsigned get(int n) {
        if (n <= 0) return 0;
        if (pos + n > nBits) { ok = false; pos = nBits; return 0; }
        const unsigned v = peek(n);
        pos += n;
        return v;
    }
};

// pacfileThem.py:219-302: per band {ba-1 | 0 : nMantSizeBits, scale factor : nScaleBits, mantissas : ba bits each or
// Huffman codes (+ ba raw bits after the escape code)}; mantissas land at the band's own lines (dense)
bool read_band_records(BitReader& r, const mrc_config& cfg, int table, const std::vector<int>& bandN, int32_t* sf,
                       int32_t* ba, int32_t* mant) {
    int line = 0;
    for (size_t band = 0; band < bandN.size(); ++band) {
        int bits = (int)r.get(cfg.n_mant_size_bits);
        if (bits) ++bits;
        if (bits > 16) return false;              // the encoder never allocates more (codecThem.py:292-293); also the
        ba[band] = bits;                          // reader's and the dequantiser's shifts are only defined up to there
        sf[band] = (int32_t)r.get(cfg.n_scale_bits);
        if (bits) {
            for (int j = 0; j < bandN[band]; ++j) {
                if (table == kRawTable) {
                    mant[line + j] = (int32_t)r.get(bits);
                } else {
                    const unsigned w = r.peek(kPeekBits);
                    const int n = kDecLut.len[table][w];
                    if (!n) return false;
                    r.get(n);
                    const int v = kDecLut.value[table][w];
                    mant[line + j] = (v == kTables[table].escape) ? (int32_t)r.get(bits) : v;
                }
            }
        }
        line += bandN[band];
    }
    return r.ok;
}

inline uint32_t get_u32le(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t get_u16le(const uint8_t* p) { return p[0] | (p[1] << 8); }

}  // namespace

namespace mrc {
void pack_tables(PackTables* out) {
    static_assert(kPackLutSize == kLutSize && kPackRawTable == kRawTable, "device and host packer tables");
    for (int t = 0; t < 4; ++t) {
        for (int v = 0; v <= kLutSize; ++v) out->emit[t * (kLutSize + 1) + v] = kLut.emit[t][v];
        out->escape[t] = kTables[t].escape;
    }
}
}  // namespace mrc

extern "C" {

int mrc_band_table(const mrc_config* cfg, int a, int b, int32_t* n_bands, int32_t* n_lines) {
    if (!shape_ok(cfg, a, b) || !n_bands) return MRC_ERR_INVALID;
    std::vector<int> cnt;
    if (!mrc::band_table(*cfg, a, b, &cnt)) return MRC_ERR_INVALID;
    *n_bands = (int32_t)cnt.size();
    if (n_lines) for (size_t i = 0; i < cnt.size(); ++i) n_lines[i] = cnt[i];
    return MRC_OK;
}

int64_t mrc_pack_bound(const mrc_config* cfg, int a, int b, int n_channels, int joint) {
    if (!shape_ok(cfg, a, b) || n_channels < 1) return MRC_ERR_INVALID;
    std::vector<int> cnt;
    if (!mrc::band_table(*cfg, a, b, &cnt)) return MRC_ERR_INVALID;
    // worst case per line: 16 raw bits + the longest escape code (6 bits); header fields on top
    const int64_t half = (a + b) / 2;
    int64_t bits = 4 + cfg->blksw_bits_a + cfg->blksw_bits_b + 4 * cfg->n_scale_bits +
                   (int64_t)cnt.size() * (1 + cfg->n_mant_size_bits + cfg->n_scale_bits) + half * (16 + 9);
    (void)joint;
    return (int64_t)n_channels * (4 + (bits + 7) / 8);
}

int mrc_pac_header(const mrc_config* cfg, int n_channels, uint32_t num_samples, uint8_t* out, int64_t out_cap,
                   int64_t* out_len) {
    if (!cfg || !out || !out_len || n_channels < 1) return MRC_ERR_INVALID;
    std::vector<int> cnt;
    if (!mrc::band_table(*cfg, cfg->n_mdct_lines, cfg->n_mdct_lines, &cnt)) return MRC_ERR_INVALID;
    const int64_t need = 4 + 4 + 2 + 4 + 4 + 2 + 2 + 4 + 2 * (int64_t)cnt.size();
    if (out_cap < need) return MRC_ERR_INVALID;
    // pacfileThem.py:595-597: padded only when numSamples ALREADY is a multiple of nMDCTLines (inverted test)
    if (num_samples % (uint32_t)cfg->n_mdct_lines == 0) num_samples += (uint32_t)cfg->n_mdct_lines;
    uint8_t* p = out;
    std::memcpy(p, "PAC ", 4); p += 4;
    put_u32le(p, (uint32_t)cfg->sample_rate); p += 4;
    p[0] = n_channels & 255; p[1] = (n_channels >> 8) & 255; p += 2;
    put_u32le(p, num_samples); p += 4;
    put_u32le(p, (uint32_t)cfg->n_mdct_lines); p += 4;
    p[0] = cfg->n_scale_bits; p[1] = 0; p += 2;
    p[0] = cfg->n_mant_size_bits; p[1] = 0; p += 2;
    put_u32le(p, (uint32_t)cnt.size()); p += 4;
    for (int c : cnt) { p[0] = c & 255; p[1] = (c >> 8) & 255; p += 2; }
    *out_len = p - out;
    return MRC_OK;
}

}  // extern "C"

// One chunk pair/set per block, exactly the bytes WriteDataBlock / JointWriteDataBlock append.
template <class MantT>
static int pack_blocks(const mrc_config* cfg, int64_t n, int nch, int a, int b, int joint, int use_huffman,
                       const int32_t* overall_scale, const int32_t* ms_switch, const int32_t* scale_factor,
                       const int32_t* bit_alloc, const MantT* mantissa, uint8_t* out, int64_t out_cap,
                       int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved,
                       const int32_t* huff_table_in = nullptr) {
    if (!shape_ok(cfg, a, b) || n < 0 || !overall_scale || !scale_factor || !bit_alloc || !mantissa || !out ||
        !block_offset || (joint && (!ms_switch || nch != 2)) || nch < 1)
        return MRC_ERR_INVALID;
    std::vector<int> nLines;
    if (!mrc::band_table(*cfg, a, b, &nLines)) return MRC_ERR_INVALID;
    const int nb = (int)nLines.size();
    const int half = (a + b) / 2;
    const int nScalePerBlock = joint ? 4 : nch;
    const uint32_t bitA = (uint32_t)(1 - a / cfg->n_mdct_lines), bitB = (uint32_t)(1 - b / cfg->n_mdct_lines);   // py2 int division
    // Two passes so that blocks can be packed by several host threads (mrc_pack_set_threads): (1) table choice
    // and chunk size of every channel chunk, (2) serial prefix sum of the sizes, (3) payloads at known offsets.
    const int64_t nChunks = n * nch;
    std::vector<ChannelPlan> plans((size_t)nChunks);
    std::vector<int64_t> chunkBytes((size_t)nChunks);
    std::atomic<int> badTable{0};
    parallel_for(nChunks, [&](int64_t c) {
        const int ch = (int)(c % nch);
        int given = -1;
        if (huff_table_in) {
            given = huff_table_in[c];
            if (given != kRawTable && (given < 0 || given > 3)) { badTable.store(1); given = kRawTable; }
        }
        const ChannelPlan plan = plan_channel(bit_alloc + c * nb, mantissa + c * (int64_t)half, nLines, use_huffman, given);
        int64_t bits = 4 + cfg->blksw_bits_a + cfg->blksw_bits_b + (int64_t)nb * (cfg->n_mant_size_bits + cfg->n_scale_bits) +
                       plan.mantBits;
        if (joint) { if (ch == 0) bits += nb + 4 * cfg->n_scale_bits; }           // pacfileThem.py:826-833
        else bits += cfg->n_scale_bits;                                           // pacfileThem.py:655
        plans[(size_t)c] = plan;
        chunkBytes[(size_t)c] = (bits + 7) / 8;                                   // pacfileThem.py:706-707
        if (huff_table) huff_table[c] = plan.table;
        if (bits_saved) bits_saved[c] = plan.bitsSaved;
    });
    if (badTable.load()) return MRC_ERR_INVALID;
    std::vector<int64_t> chunkPos((size_t)nChunks + 1);
    int64_t pos = 0;
    for (int64_t c = 0; c < nChunks; ++c) {
        if (c % nch == 0) block_offset[c / nch] = pos;
        chunkPos[(size_t)c] = pos;
        pos += 4 + chunkBytes[(size_t)c];
    }
    if (pos > out_cap) return MRC_ERR_NOMEM;
    std::atomic<int> bad{0};
    parallel_for(nChunks, [&](int64_t c) {
        const int64_t i = c / nch;
        const int ch = (int)(c % nch);
        const ChannelPlan& plan = plans[(size_t)c];
        uint8_t* dst = out + chunkPos[(size_t)c];
        put_u32le(dst, (uint32_t)chunkBytes[(size_t)c]);
        BitWriter w(dst + 4);
        w.put((uint32_t)plan.table, 4);
        w.put(bitA, cfg->blksw_bits_a);
        w.put(bitB, cfg->blksw_bits_b);
        if (joint) {
            if (ch == 0) {
                for (int s4 = 0; s4 < 4; ++s4) w.put((uint32_t)overall_scale[i * 4 + s4], cfg->n_scale_bits);   // L,R,M,S
                for (int k = 0; k < nb; ++k) w.put((uint32_t)ms_switch[i * nb + k], 1);
            }
        } else {
            w.put((uint32_t)overall_scale[i * nScalePerBlock + ch], cfg->n_scale_bits);
        }
        write_band_records(w, *cfg, scale_factor + c * nb, bit_alloc + c * nb, mantissa + c * (int64_t)half, nLines, plan.table);
        w.flush();
        if (w.p - (dst + 4) != chunkBytes[(size_t)c]) bad.store(1);              // size law and writer disagree: bug
    });
    if (bad.load()) return MRC_ERR_INVALID;
    block_offset[n] = pos;
    return MRC_OK;
}

extern "C" {

int mrc_pack_set_threads(int n_threads) {
    if (n_threads < 1 || n_threads > 1024) return MRC_ERR_INVALID;
    g_packThreads.store(n_threads);
    return MRC_OK;
}

int mrc_pack_blocks(const mrc_config* cfg, int64_t n_blocks, int n_channels, int a, int b, int use_huffman,
                    const int32_t* overall_scale, const int32_t* scale_factor, const int32_t* bit_alloc,
                    const int32_t* mantissa, uint8_t* out, int64_t out_cap, int64_t* block_offset,
                    int32_t* huff_table, int32_t* bits_saved) {
    return pack_blocks(cfg, n_blocks, n_channels, a, b, 0, use_huffman, overall_scale, nullptr, scale_factor, bit_alloc,
                       mantissa, out, out_cap, block_offset, huff_table, bits_saved);
}

int mrc_pack_joint_blocks(const mrc_config* cfg, int64_t n_blocks, int a, int b, int use_huffman,
                          const int32_t* overall_scale, const int32_t* ms_switch, const int32_t* scale_factor,
                          const int32_t* bit_alloc, const int32_t* mantissa, uint8_t* out, int64_t out_cap,
                          int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved) {
    return pack_blocks(cfg, n_blocks, 2, a, b, 1, use_huffman, overall_scale, ms_switch, scale_factor, bit_alloc,
                       mantissa, out, out_cap, block_offset, huff_table, bits_saved);
}

int mrc_pack_blocks_with_tables(const mrc_config* cfg, int64_t n_blocks, int n_channels, int a, int b,
                                const int32_t* huff_table_in, const int32_t* overall_scale, const int32_t* scale_factor,
                                const int32_t* bit_alloc, const int32_t* mantissa, uint8_t* out, int64_t out_cap,
                                int64_t* block_offset) {
    if (!huff_table_in) return MRC_ERR_INVALID;
    return pack_blocks(cfg, n_blocks, n_channels, a, b, 0, 1, overall_scale, nullptr, scale_factor, bit_alloc, mantissa,
                       out, out_cap, block_offset, nullptr, nullptr, huff_table_in);
}

int mrc_pack_joint_blocks_with_tables(const mrc_config* cfg, int64_t n_blocks, int a, int b, const int32_t* huff_table_in,
                                      const int32_t* overall_scale, const int32_t* ms_switch,
                                      const int32_t* scale_factor, const int32_t* bit_alloc, const int32_t* mantissa,
                                      uint8_t* out, int64_t out_cap, int64_t* block_offset) {
    if (!huff_table_in) return MRC_ERR_INVALID;
    return pack_blocks(cfg, n_blocks, 2, a, b, 1, 1, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa, out,
                       out_cap, block_offset, nullptr, nullptr, huff_table_in);
}

int mrc_pack_get_threads(void) { return g_packThreads.load(); }

int mrc_pack_blocks_ex(const mrc_config* cfg, int64_t n_blocks, int n_channels, int a, int b, int joint, int use_huffman,
                       const int32_t* huff_table_in, const int32_t* overall_scale, const int32_t* ms_switch,
                       const int32_t* scale_factor, const int32_t* bit_alloc, const void* mantissa, int mantissa_format,
                       uint8_t* out, int64_t out_cap, int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved) {
    if (joint && n_channels != 2) return MRC_ERR_INVALID;
    if (mantissa_format == MRC_MANTISSA_I16)
        return pack_blocks(cfg, n_blocks, n_channels, a, b, joint ? 1 : 0, use_huffman, overall_scale, ms_switch,
                           scale_factor, bit_alloc, (const uint16_t*)mantissa, out, out_cap, block_offset, huff_table,
                           bits_saved, huff_table_in);
    if (mantissa_format == MRC_MANTISSA_I32)
        return pack_blocks(cfg, n_blocks, n_channels, a, b, joint ? 1 : 0, use_huffman, overall_scale, ms_switch,
                           scale_factor, bit_alloc, (const int32_t*)mantissa, out, out_cap, block_offset, huff_table,
                           bits_saved, huff_table_in);
    return MRC_ERR_INVALID;
}


// ---- decode side ---------------------------------------------------------------------------------------------
int mrc_pac_read_header(const uint8_t* buf, int64_t len, mrc_config* cfg, int32_t* n_channels, uint32_t* num_samples,
                        int64_t* data_offset) {
    if (!buf || !cfg || !n_channels || !num_samples || !data_offset || len < 26) return MRC_ERR_INVALID;
    if (std::memcmp(buf, "PAC ", 4) != 0) return MRC_ERR_INVALID;
    cfg->sample_rate = (int32_t)get_u32le(buf + 4);
    *n_channels = (int32_t)get_u16le(buf + 8);
    *num_samples = get_u32le(buf + 10);
    cfg->n_mdct_lines = (int32_t)get_u32le(buf + 14);
    cfg->n_scale_bits = (int32_t)get_u16le(buf + 18);
    cfg->n_mant_size_bits = (int32_t)get_u16le(buf + 20);
    const uint32_t nBands = get_u32le(buf + 22);
    if (nBands > 4096 || 26 + 2 * (int64_t)nBands > len) return MRC_ERR_INVALID;
    // the header is untrusted input and sizes every later allocation: refuse what no encoder writes
    if (cfg->sample_rate <= 0 || *n_channels < 1 || *n_channels > 2 || cfg->n_mdct_lines < 16 ||
        cfg->n_mdct_lines > 8192 || (cfg->n_mdct_lines & (cfg->n_mdct_lines - 1)) != 0 || cfg->n_scale_bits < 1 ||
        cfg->n_scale_bits > 4 || cfg->n_mant_size_bits < 1 || cfg->n_mant_size_bits > 8)
        return MRC_ERR_INVALID;
    *data_offset = 26 + 2 * (int64_t)nBands;           // the band table itself is implied by rate and block length
    return MRC_OK;
}

int64_t mrc_pac_scan_chunks(const uint8_t* buf, int64_t len, int64_t data_offset, int64_t* chunk_offset, int64_t cap) {
    if (!buf || data_offset < 0 || data_offset > len) return MRC_ERR_INVALID;
    int64_t n = 0, off = data_offset;
    while (off + 4 <= len) {
        const int64_t nBytes = get_u32le(buf + off);
        if (off + 4 + nBytes > len) return MRC_ERR_INVALID;        // truncated chunk
        if (chunk_offset && n < cap) chunk_offset[n] = off;
        ++n;
        off += 4 + nBytes;
    }
    return n;
}

int mrc_unpack_blocks(const mrc_config* cfg, int64_t n_blocks, int n_channels, int joint, const uint8_t* buf, int64_t len,
                      const int64_t* chunk_offset, int32_t* a_out, int32_t* b_out, int32_t* huff_table,
                      int32_t* overall_scale, int32_t* ms_switch, int32_t* scale_factor, int32_t* bit_alloc,
                      int32_t* mantissa) {
    if (!cfg || !buf || !chunk_offset || !a_out || !b_out || !huff_table || !overall_scale || !scale_factor ||
        !bit_alloc || !mantissa || n_blocks < 0 || n_channels < 1 || n_channels > 2 || (joint && (n_channels != 2 || !ms_switch)) ||
        !shape_ok(cfg, cfg->n_mdct_lines, cfg->n_mdct_lines))
        return MRC_ERR_INVALID;
    const int L = cfg->n_mdct_lines, nOs = joint ? 4 : n_channels;
    std::atomic<int> bad{0};
    parallel_for(n_blocks, [&](int64_t blk) {
        std::vector<int> bandN;
        for (int ch = 0; ch < n_channels; ++ch) {
            const int64_t off = chunk_offset[blk * n_channels + ch];
            if (off < 0 || off + 4 > len) { bad = 1; return; }
            const int64_t nBytes = get_u32le(buf + off);
            if (off + 4 + nBytes > len) { bad = 1; return; }
            BitReader r(buf + off + 4, nBytes);
            const int table = (int)r.get(4);
            if (table != kRawTable && table > 3) { bad = 1; return; }
            const int swA = (int)r.get(cfg->blksw_bits_a), swB = (int)r.get(cfg->blksw_bits_b);
            const int a = swA ? cfg->n_short : L, b = swB ? cfg->n_short : L;      // pacfileThem.py:206-207
            if (ch == 0) { a_out[blk] = a; b_out[blk] = b; }
            else if (a != a_out[blk] || b != b_out[blk]) { bad = 1; return; }
            if (!mrc::band_table(*cfg, a, b, &bandN) || (int)bandN.size() > MRC_MAX_BANDS) { bad = 1; return; }
            huff_table[blk * n_channels + ch] = table;
            if (joint) {
                if (ch == 0) {
                    for (int i = 0; i < 4; ++i) overall_scale[blk * 4 + i] = (int32_t)r.get(cfg->n_scale_bits);
                    for (int i = 0; i < MRC_MAX_BANDS; ++i)
                        ms_switch[blk * MRC_MAX_BANDS + i] = i < (int)bandN.size() ? (int32_t)r.get(1) : 0;
                }
            } else {
                overall_scale[blk * nOs + ch] = (int32_t)r.get(cfg->n_scale_bits);
            }
            int32_t* sf = scale_factor + (blk * n_channels + ch) * MRC_MAX_BANDS;
            int32_t* ba = bit_alloc + (blk * n_channels + ch) * MRC_MAX_BANDS;
            int32_t* m = mantissa + (blk * n_channels + ch) * (int64_t)L;
            std::memset(sf, 0, sizeof(int32_t) * MRC_MAX_BANDS);
            std::memset(ba, 0, sizeof(int32_t) * MRC_MAX_BANDS);
            std::memset(m, 0, sizeof(int32_t) * L);
            if (!read_band_records(r, *cfg, table, bandN, sf, ba, m)) { bad = 1; return; }
        }
    });
    return bad.load() ? MRC_ERR_INVALID : MRC_OK;
}

}  // extern "C"
