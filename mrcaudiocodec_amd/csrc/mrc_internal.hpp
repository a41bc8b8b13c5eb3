// Internal declarations shared by the host glue (mrc_api.cpp, mrc_tables.cpp) and the gfx950
// kernels (mrc_kernels.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "mrc_hip.h"

namespace mrc {

constexpr int kMaxRadices = 8;
// how a channel's samples are held: float64 signed fractions (what pcmfile.py:98 hands the codec) or the file's
// int16 PCM codes, converted on load (dev::pcm16_to_frac)
constexpr int kSampleF64 = MRC_SAMPLES_F64, kSampleI16 = MRC_SAMPLES_PCM16;
constexpr int kMaxBands = MRC_MAX_BANDS;

// Everything a kernel needs to know about one block shape (a,b).  POD, passed by value; the
// pointers address one device blob owned by the handle.
struct LineConstants { double z, quiet, lowE; int band, pad; };      // 32 bytes: two 16-byte loads per line

struct DevShape {
    int a, b, N, halfN, Q, H;        // N = a+b, halfN = N/2 lines, Q = N/4 (MDCT FFT), H = N/2 (psycho FFT)
    int shift;                       // (b-a)/4: signed circular shift that maps n0=(b+1)/2 to the standard phase
    int nBands;
    int peakLast;                    // N/2 - 100 (psychoac.py:160)
    int nRadQ, nRadH;
    int radQ[kMaxRadices], radH[kMaxRadices];
    int nScaleBits, maxMantBits;
    double twoOverN;                 // 2.0/N (mdct.py:76)
    double binHz;                    // py2 integer sampleRate/N (psychoac.py:165)
    double xiDen;                    // (N**2.)*(3./8.) (psychoac.py:151)
    double budgetMono;               // codecThem.py:299-306 (reservoir added last)
    double budgetJointPre;           // codecThem.py:381-388 (before `+= bitReservoir`)
    double blkswA, blkswB;
    const double* win;               // [N] transition window (window.py:104-121)
    int winSymmetric;                // win[n] == win[N - 1 - n] bit for bit (true for a = b)
    const double* hann;              // [N] window.py:28-45
    const double2* pre;              // [Q] exp(-i pi (4n+1)/(4M)), M = N/2
    const double2* post;             // [Q] exp(-i pi k / M)
    const double2* wQ;               // [Q] exp(-2 pi i t/Q)
    const double2* wH;               // [H] exp(-2 pi i t/H)
    const double2* wN;               // [H] exp(-2 pi i k/N)
    const double2* fftTw;            // H = 1024 only: per-pass twiddles of the psycho FFT (dev::fft_regs_1024), else null
    const double* zb;                // [halfN] Bark(MDCTFreq) (psychoac.py:27-29,142-143)
    const double* quiet;             // [halfN] Intensity(Thresh(MDCTFreq)) (psychoac.py:155)
    const double* lowE;              // [halfN] 2^(2.7 log2(10) (zb+1/2)): per-line factor of the -27 dB/Bark lower slope
    const int* bandLo;               // [nBands]
    const int* bandN;                // [nBands]
    const unsigned char* bandOfLine; // [halfN]
    const struct LineConstants* lineC;  // [halfN] zb, quiet, lowE and the band of a line side by side (smr_kernel's sweep)
    const unsigned short* loLine;    // [halfN] first line j with zb[j] - zb[k] >= -1/2 (search hint)
    const unsigned short* hiLine;    // [halfN] first line j with zb[j] - zb[k] > 1/2, halfN if none (search hint)
    double linesPerHz;               // N / sampleRate
    // NumPy's pairwise summation of every band's lines as a static tree (ms_plan): msLeaves (lo, n) runs of <= 128
    // lines, then internal nodes (left, right) in an order where children come first, then the root node of each band
    const int* msPlan;               // [2 msLeaves + 2 msInternal + nBands]
    int msLeaves, msInternal;
};

struct HostShape {
    DevShape dev{};                  // device view (pointers valid on the device)
    std::vector<int> bandN, bandLo;  // host copies
    void* blob = nullptr;            // device allocation backing dev.*
};

// mrc_tables.cpp
bool band_table(const mrc_config& cfg, int a, int b, std::vector<int>* count);   // host only
// The summation tree np.sum walks over each band's contiguous run of lines (pairwise summation: runs of more than 128
// elements are halved, the first half rounded down to a multiple of 8).  -> plan laid out as DevShape::msPlan.
void ms_plan(const std::vector<int>& bandLo, const std::vector<int>& bandN, std::vector<int>* plan, int* nLeaves,
             int* nInternal);
bool build_shape(const mrc_config& cfg, int a, int b, HostShape* out, std::string* err);
void free_shape(HostShape* s);
int scale_factor_host(double v, int nScaleBits, int nMantBits);

// mrc_kernels.hip -- launchers (enqueue only)
hipError_t launch_mdct(const DevShape& S, int64_t nFrames, const void* chL, const void* chR, int fmt,
                       int64_t stride, const int64_t* offsets, bool applyWindow, double* lines, int* oscale,
                       hipStream_t st);
// mrc_kernels_long.hip -- long-block specialisation (a = b = 1024)
bool mdct_long_applicable(const DevShape& S, int64_t stride, const int64_t* offsets, const void* chL,
                          const void* chR, int fmt);
hipError_t launch_mdct_long(const DevShape& S, int64_t nFrames, const void* chL, const void* chR, int fmt,
                            int64_t stride, const int64_t* offsets, double* lines, int* oscale, hipStream_t st);
hipError_t launch_window(const DevShape& S, int64_t nBlocks, const double* in, double* out, hipStream_t st);
hipError_t launch_unscale(int64_t nBlocks, int halfN, const double* scaled, const int* oscale, double* lines,
                          hipStream_t st);
hipError_t launch_smr(const DevShape& S, int64_t nFrames, const void* chL, const void* chR, int fmt,
                      int64_t stride, const int64_t* offsets, const double* lines, const int* oscale,
                      double* smr, double* thresh, double* bandPeak /* [frames*signals][nBands] or null */,
                      const int* msSwitch /* joint: [frames][nBands] -> only the SMRs the encoder uses are computed; null: all */,
                      bool exactSpread, hipStream_t st,
                      unsigned long long* sens = nullptr /* MRC_OPT_SENSITIVITY: counters [MRC_SENS_COUNT] on the device */);
// mrc_kernels_sens.hip: decisions within a guard band of rounding (quantiser edges, allocation ties, M/S threshold)
hipError_t launch_sensitivity(const DevShape& S, int64_t nFrames, int joint, const double* lines, const int* oscale,
                              const double* smr, const double* bandPeak, const int* msSwitch, const int* bitAlloc,
                              const int* scaleFactor, unsigned long long* sens, unsigned char* frameFlags, hipStream_t st);
hipError_t launch_alloc_quant(const DevShape& S, int64_t nFrames, int joint, const double* lines,
                              const int* oscale, const double* smr, const int* resIn, int* msSwitch,
                              int* bitAlloc, int* scaleFactor, void* mantissa, int mantFmt /* MRC_MANTISSA_* */,
                              int* resOut, double* bandPeakWs,
                              bool peaksReady /* bandPeakWs already filled by launch_smr */,
                              bool msReady /* msSwitch already filled (launch_ms_switch ran before launch_smr) */,
                              hipEvent_t* ev /* null or 2 events: after band_stats, after bitalloc */, hipStream_t st);
hipError_t launch_pcm_to_float(int64_t n, const short* pcm, double* out, hipStream_t st);
hipError_t launch_quantize_uniform(int64_t n, int nBits, const double* x, long long* out, hipStream_t st);
hipError_t launch_bark(int64_t n, const double* f, double* out, hipStream_t st);
size_t alloc_workspace_bytes(const DevShape& S, int64_t nFrames, int joint);   // bandPeakWs size
// mrc_kernels_decode.hip
hipError_t launch_decode(const DevShape& S, int64_t nBlocks, int nStreams, const int* oscale, const int* msSwitch,
                         const int* scaleFactor, const int* bitAlloc, const int* mantissa, const int64_t* outOffset,
                         double* outL, double* outR, hipStream_t st);
hipError_t launch_pcm16(int64_t n, const double* x, short* out, hipStream_t st);
// mrc_kernels_pack.hip -- `.pac` chunk packing on the device
constexpr int kPackLutSize = 65;     // the largest value in any Huffman table is 64
constexpr int kPackRawTable = 15;    // codecThem.py:149
struct PackTables {                  // mrc_pack.cpp: pack_tables()
    // per table and value (last index: any other value): code | length << 16 | 1 << 31 where the raw mantissa follows
    unsigned emit[4 * (kPackLutSize + 1)];
    int escape[4];
};
struct PackParams {
    int nch, joint, useHuffman;
    int nScaleBits, nMantSizeBits, blkBitsA, blkBitsB;      // field widths
    unsigned bitA, bitB;                                    // block-switching bits of this shape (pacfileThem.py:720-723)
};
void pack_tables(PackTables* out);                          // host, from the table data in mrc_pack.cpp
size_t pack_workspace_bytes(int64_t nChunks);
hipError_t launch_pack(const DevShape& S, const PackParams& P, const PackTables& T, int64_t nBlocks, const int* oscale,
                       const int* msSwitch, const int* scaleFactor, const int* bitAlloc, const void* mant, int mantFmt,
                       const int* tableIn /* nullable */, int* tableOut, int* bitsSaved /* nullable */, unsigned char* out,
                       long long outCap, long long* blockOffset /* [nBlocks + 1] */, void* ws /* pack_workspace_bytes */,
                       int boundBytes /* largest chunk payload */, bool allBandsNonEmpty /* of this shape's table */,
                       hipStream_t st);
// The three steps of launch_pack on their own, for chunks that several shape groups contribute to ONE output in a
// given order (the chained stream encode): chunkMap [nBlocks * nch] (nullable) = where each of this group's chunks sits
// in the global chunk order; chunkBytes / pos are indexed by that global index.  chunkStream (nullable) [nChunks]:
// the stream each global chunk belongs to -- every stream's first chunk is preceded by a header of hdrLen bytes.
struct PackWs {                      // views into a workspace of pack_workspace_bytes(nChunks)
    long long* pos; long long* tileSum; long long* total; int* errorFlag; int* chunkBytes;
};
PackWs pack_ws_views(void* ws, int64_t nChunks);
hipError_t launch_pack_plan(const DevShape& S, const PackParams& P, const PackTables& T, int64_t nBlocks,
                            const int* bitAlloc, const void* mant, int mantFmt, const int* tableIn, int* tableOut,
                            int* bitsSaved, const PackWs& W, const long long* chunkMap, bool allBandsNonEmpty,
                            hipStream_t st);
hipError_t launch_pack_scan(int64_t nChunks, int nch /* 0: no block offsets */, const PackWs& W, long long* blockOffset,
                            const int* chunkStream, int hdrLen, hipStream_t st);
hipError_t launch_pack_write(const DevShape& S, const PackParams& P, const PackTables& T, int64_t nBlocks,
                             const int* oscale, const int* msSwitch, const int* scaleFactor, const int* bitAlloc,
                             const void* mant, int mantFmt, const int* table, const PackWs& W, const long long* chunkMap,
                             unsigned char* out, long long outCap, int boundBytes, bool allBandsNonEmpty, hipStream_t st);
hipError_t launch_pack_export(const void* ws, int64_t nChunks, long long* hostOut /* page-locked: {total, error flag} */,
                              hipStream_t st);
const int* pack_error_flag(const void* ws, int64_t nChunks);          // device addresses inside ws
const long long* pack_total_bytes(const void* ws, int64_t nChunks);
// mrc_kernels_chain.hip -- chained stream encode: reservoir-free preparation per block, serial scan per stream
struct ChainGroupDev {               // what chain_phase_b_kernel knows about one block-shape group (device memory)
    int joint, nb, nTot, M, K, nEv, maxN, nScaleBits, nstream, pad_;
    double budgetMono, budgetJointPre, blkswA, blkswB;          // codecThem.py:299-308, 381-396
    const unsigned char* bandOfLine;
    const int* bandN;
    const double* lines;             // [n][nsig][M] unscaled MDCT lines (phase A)
    const double* peak;              // [n][nsig][nb] per-band max |X| of the unscaled lines
    const int* oscale;               // [n][nsig] overall scales
    const int* ms;                   // [n][nb] M/S switch (joint groups)
    const unsigned* ev;              // [n][nEv] grant events in np.argmax's order: band | bitsAfter << 6 | nLines << 11
    const unsigned* pre;             // [n][nEv + 1] bits spent before each event if all before it are granted
    int* bitAlloc;                   // [n][nstream][nb]
    int* scaleFactor;                // [n][nstream][nb]
    unsigned short* mant;            // [n][nstream][M]
    int* table;                      // [n][nstream] Huffman table id (15 = raw)
};
size_t chain_events_per_block(const DevShape& S, int joint);
hipError_t launch_chain_prep(const DevShape& S, int joint, int64_t nBlocks, const double* smr, const int* msSwitch,
                             unsigned* ev, unsigned* pre, int forceFallback, hipStream_t st);
hipError_t launch_chain_phase_b(int64_t nStreams, const ChainGroupDev* groups, const int* items, const long long* itemStart,
                                int* reservoir, int* resTrace, int useHuffman, int threads /* 0: chosen by stream count */,
                                hipStream_t st);
hipError_t launch_chain_flush_gather(int64_t nStreams, int L, const void* pcmL, const void* pcmR, int fmt, int64_t stride,
                                     const long long* tailOffset, void* out, hipStream_t st);
hipError_t launch_chain_headers(int64_t nStreams, int hdrLen, const unsigned char* hdr, const long long* firstChunk,
                                const long long* pos, unsigned char* out, long long outCap, long long* streamPos,
                                hipStream_t st);
// mrc_kernels_huff.hip
hipError_t launch_huffman_gain(const DevShape& S, int64_t nFrames, int nStreams, const int* bitAlloc,
                               const int* mantissa, const int* reservoirOut, int* huffTable, int* bitsSaved,
                               int* reservoirNext, hipStream_t st);
hipError_t launch_bitalloc_cases(int64_t nCases, int nBands, int maxMantBits, const int* nLines,
                                 const double* budget, const double* smr, int* bits, int* left,
                                 double* smrAfter /* nullable: running SMRs after the loop */, hipStream_t st);
hipError_t launch_scale_factor(int64_t n, int nScaleBits, const double* v, const int* nMantBits, int* out,
                               hipStream_t st);
hipError_t launch_mantissa(int64_t n, int nScaleBits, const double* x, const int* scale, const int* nMantBits,
                           int* out, hipStream_t st);
hipError_t launch_transient_peaks(int64_t nHops, int nCh, int hop, int nShort, int nSec, const double* sos, bool unitB0,
                                  const void* streams, int fmt, int64_t chStride, double* peaks, hipStream_t st);
hipError_t launch_stereo_masking(int64_t n, const double* mid, const double* side, const double* z, double* outMid,
                                 double* outSide, hipStream_t st);
hipError_t launch_ms_switch(int64_t nBlocks, int nBands, int nLeaves, int nInternal, const int* plan /* device */,
                            const double* L, const double* R, int64_t blockStride /* doubles between blocks */,
                            int nLines /* lines per block */, int* out, hipStream_t st);

}  // namespace mrc
