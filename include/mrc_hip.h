/*
 * mrc_hip.h -- C ABI of libmrc_hip.so: the MI355X (gfx950) implementation of the per-block ENCODE
 * hot path of laser55/mrcAudioCodec (KBD/transition window -> MDCT -> FFT psychoacoustic masked
 * threshold / SMR -> greedy bit allocation -> scale-factor / mantissa quantise -> per-band M/S).
 *
 * The reference has no FFI of its own: its plugin seam is the Python module `codecThem`
 * (pacfileThem.py:108 `import codecThem as codec`; calls at pacfileThem.py:649 -> 987-994 and
 * 820 -> 996-1003).  The entry points below are what a binding for that seam needs; the drop-in
 * Python module mrcaudiocodec_amd/codecThem.py binds them with ctypes.
 *
 * Conventions: every function returns 0 on success, a negative mrc_status on error
 * (mrc_last_error(h) gives the text); no exceptions cross the ABI; a handle is bound to one HIP
 * device and calls on one handle must be serialised by the caller.  There is NO CPU fallback: if
 * no gfx950 device is usable, mrc_create fails with MRC_ERR_NO_DEVICE.
 * "host" functions take host pointers (C-contiguous, caller-allocated outputs) and copy for the
 * caller; "dev" functions take device pointers (HIP allocations of the handle's device) and enqueue
 * on the given hipStream_t (passed as void*; NULL = the handle's own stream) without synchronising.
 *
 * Block shapes.  A block is `a` samples carried over from the previous call followed by `b` new
 * samples (pacfileThem.py:628-631), N = a+b, N/2 MDCT lines.  With nMDCTLines = 1024 and
 * nSamplesShort = 128 the shapes are (1024,1024) long / 25 bands, (128,128) short / 9 bands and the
 * two transitions (1024,128), (128,1024) with 576 lines / 9 bands (pacfileThem.py:637-645,1192-1210).
 *
 * Dense output layout (per block): overall_scale[nsig], scale_factor[nstream][nBands],
 * bit_alloc[nstream][nBands], mantissa[nstream][N/2] (0 where the band got no bits; the reference
 * omits those bands, codecThem.py:336-350 -- the Python wrapper compacts), reservoir_out.
 * mono: nsig = nstream = 1.  joint stereo: nsig = 4 in the order L,R,M,S (codecThem.py:456-460),
 * nstream = 2 (stream 0 = Mid-or-Left per band, stream 1 = Side-or-Right, codecThem.py:524-551),
 * plus ms_switch[nBands].
 */
#ifndef MRC_HIP_H
#define MRC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRC_VERSION 300            /* 0.3.0 */
#define MRC_MAX_BANDS 32
/* how a channel's samples are held: float64 signed fractions (what pcmfile.py:98 hands the codec) or the file's
 * int16 PCM codes (converted on load, pcmfile.py:91-100); how the mantissa plane is stored */
#define MRC_SAMPLES_F64 0
#define MRC_SAMPLES_PCM16 1
#define MRC_MANTISSA_I32 0
#define MRC_MANTISSA_I16 1         /* uint16: a code is sign bit + magnitude in at most 16 bits (codecThem.py:292-293) */

typedef enum mrc_status {
    MRC_OK = 0,
    MRC_ERR_INVALID = -1,          /* bad argument (shape, sizes, null pointer) */
    MRC_ERR_NO_DEVICE = -2,        /* no usable HIP device / not gfx950 */
    MRC_ERR_HIP = -3,              /* a HIP runtime call failed */
    MRC_ERR_NOMEM = -4
} mrc_status;

/* The codec parameters the path reads from the reference's CodingParams bag
 * (audiofile.py:51-53; values set at pacfileThem.py:1105-1121). */
typedef struct mrc_config {
    int32_t sample_rate;           /* codingParams.sampleRate (integer Hz: py2 `sampleRate/N` is an int division) */
    int32_t n_mdct_lines;          /* codingParams.nMDCTLines, 1024 */
    int32_t n_short;               /* codingParams.nSamplesShort, 128 */
    int32_t n_scale_bits;          /* codingParams.nScaleBits, 4 */
    int32_t n_mant_size_bits;      /* codingParams.nMantSizeBits, 4 */
    int32_t blksw_bits_a;          /* codingParams.blkswBitA, 1 */
    int32_t blksw_bits_b;          /* codingParams.blkswBitB, 1 */
    int32_t device_id;             /* HIP device ordinal */
    double  target_bits_per_sample;/* codingParams.targetBitsPerSample, 2.86 */
} mrc_config;

typedef struct mrc_handle mrc_handle;

int  mrc_version(void);
int  mrc_device_count(void);                              /* number of HIP devices, 0 if none */
void mrc_default_config(mrc_config* cfg);                 /* pacfileThem.py:1105-1121 defaults, 48 kHz, device 0 */
int  mrc_create(const mrc_config* cfg, mrc_handle** out);
void mrc_destroy(mrc_handle* h);
const char* mrc_last_error(const mrc_handle* h);          /* h may be NULL: error of the last failed mrc_create */

/* Shape queries (band tables of psychoac.py:86-131 / pacfileThem.py:637-645). */
int  mrc_shape_bands(mrc_handle* h, int a, int b, int32_t* n_bands, int32_t* n_lines /*[MRC_MAX_BANDS]*/);
int  mrc_shape_budget(mrc_handle* h, int a, int b, int joint, int32_t reservoir, double* bit_budget);

/* ---- host entry points: what codecThem.EncodeSingleChannel / JointEncodeChannels bind ---------- */

/* codecThem.py:281-354 for n_blocks independent blocks of one shape.  blocks: [n_blocks][a+b].
 * reservoir_in may be NULL (zeros).  mdct_out (unscaled lines, [n_blocks][N/2]) may be NULL. */
int mrc_encode_mono(mrc_handle* h, int64_t n_blocks, int a, int b, const double* blocks,
                    const int32_t* reservoir_in,
                    int32_t* overall_scale, int32_t* scale_factor, int32_t* bit_alloc,
                    int32_t* mantissa, int32_t* reservoir_out, double* mdct_out);

/* codecThem.py:359-574.  left/right: [n_blocks][a+b].  overall_scale: [n_blocks][4] (L,R,M,S);
 * ms_switch: [n_blocks][nBands]; scale_factor/bit_alloc: [n_blocks][2][nBands];
 * mantissa: [n_blocks][2][N/2]; mdct_out: [n_blocks][4][N/2] or NULL. */
int mrc_encode_joint(mrc_handle* h, int64_t n_blocks, int a, int b, const double* left, const double* right,
                     const int32_t* reservoir_in,
                     int32_t* overall_scale, int32_t* ms_switch, int32_t* scale_factor, int32_t* bit_alloc,
                     int32_t* mantissa, int32_t* reservoir_out, double* mdct_out);

/* The same for n_blocks blocks of MIXED shapes in one call -- a block-switched stream as pacfileThem.py:1192-1210
 * produces it -- so that a binding without array libraries can hand over a whole stream: `blocks` (left / right) is
 * the blocks packed back to back, block i holding a[i] + b[i] samples; SURVEY.md 8(b) `a[]`, `b[]`.  Blocks are grouped
 * by shape inside (one launch set per distinct shape); reservoir_in [n] may be NULL.  Because the shape varies, the
 * outputs have fixed strides: scale_factor / bit_alloc [n][streams][MRC_MAX_BANDS], ms_switch [n][MRC_MAX_BANDS],
 * mantissa [n][streams][n_mdct_lines] dense (entries beyond a block's band / line count are 0), overall_scale
 * [n] (mono) / [n][4] (joint). */
int mrc_encode_mono_blocks(mrc_handle* h, int64_t n_blocks, const double* blocks, const int32_t* a, const int32_t* b,
                           const int32_t* reservoir_in, int32_t* overall_scale, int32_t* scale_factor,
                           int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out);
int mrc_encode_joint_blocks(mrc_handle* h, int64_t n_blocks, const double* left, const double* right, const int32_t* a,
                            const int32_t* b, const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch,
                            int32_t* scale_factor, int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out);

/* ---- the file's own sample format in, compact codes out: what a WAV -> .pac pipeline moves over PCIe ----------
 * A long-block stream from 16-bit PCM codes in HOST memory to codes in HOST memory, pipelined: pcm_left (and
 * pcm_right: joint stereo) hold (n_frames + 1) * n_mdct_lines int16 samples, hop-overlapped (frame f = samples
 * [f * L, f * L + 2 L); the first hop is the prior block, zeros at file start: pacfileThem.py:615-631).  The codes
 * are converted on load exactly as pcmfile.py:91-100 does (x = 2c/65535 correctly rounded, -32768 -> 0.0).
 * Frames are independent given reservoir_in [n] (NULL = zeros).  Outputs as mrc_encode_mono / _joint for a = b = L,
 * except the mantissa plane, which is uint16 [n][streams][L] (codes are at most 16 bits wide, codecThem.py:292-293).
 * The work is cut into chunks of chunk_frames frames (0: 32 768, fewer for streams shorter than six chunks) over a
 * ring of four chunk buffers; ALL uploads are queued on one HIP stream, all kernels on the handle's stream, all
 * downloads on a third, ordered by events, so chunk i+1 is copied in and chunk i-1 copied out beside chunk i's
 * kernels (one stream per direction: what the copy engines of the MI355X box run fastest).  With host buffers from
 * mrc_host_alloc / mrc_host_register (page-locked) the copies are truly asynchronous; pageable buffers work, slower.
 * PCIe bytes per (frame, channel): 2 048 in, 2 048 + 2 * 4 * nBands + 8 (+ 4 * nBands per joint frame) out. */
int mrc_encode_stream_pcm16(mrc_handle* h, int64_t n_frames, const int16_t* pcm_left, const int16_t* pcm_right,
                            const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch,
                            int32_t* scale_factor, int32_t* bit_alloc, uint16_t* mantissa16, int32_t* reservoir_out,
                            int64_t chunk_frames);
/* The same pipeline with the back end on the device too (mrc_dev_pack_blocks behind the kernels of every chunk):
 * 16-bit PCM in host memory -> the `.pac` CHUNK BYTES of the frames in host memory, what WriteDataBlock (mono:
 * one chunk per frame) / JointWriteDataBlock (stereo: two) would append frame after frame (pacfileThem.py:652-781,
 * 825-963) -- a WAV -> .pac pipeline is then: this call, file header (mrc_pac_header), write.  Only ~350 bytes per
 * frame and channel come back over PCIe instead of the 2 KB mantissa plane.  Frames independent given
 * reservoir_in [n] (NULL = zeros); use_huffman: price the four tables per chunk (codecThem.py:136-203), else raw.
 * out [out_cap] (size it with n * channels * mrc_pack_bound(cfg, L, L, 1, joint), or less and retry on
 * MRC_ERR_NOMEM), block_offset [n + 1] byte offset of every frame's chunks, total_bytes: the sum; huff_table /
 * bits_saved [n][channels] and reservoir_out [n] may be NULL.  All pointers HOST memory, page-locked for speed. */
int mrc_encode_stream_pcm16_pac(mrc_handle* h, int64_t n_frames, const int16_t* pcm_left, const int16_t* pcm_right,
                                const int32_t* reservoir_in, int use_huffman, uint8_t* out, int64_t out_cap,
                                int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved, int32_t* reservoir_out,
                                int64_t* total_bytes, int64_t chunk_frames);
/* Page-locked host memory (hipHostMalloc / hipHostRegister): no handle needed, any thread. */
int mrc_host_alloc(void** out, size_t bytes);
int mrc_host_free(void* p);
int mrc_host_register(void* p, size_t bytes);
int mrc_host_unregister(void* p);

/* ---- stage-level host entry points (parity tests against the reference's own modules) ---------- */

/* window.py:104-121 (TransitionWindow; KBDWindow when a == b): out[n_blocks][a+b] = blocks * window. */
int mrc_window(mrc_handle* h, int64_t n_blocks, int a, int b, const double* blocks, double* out);
/* mdct.py:63-76 (+ window.py:104-121 when apply_window != 0) + codecThem.py:321-322: MDCT lines
 * (unscaled, [n_blocks][N/2]) and overall scale of each block. */
int mrc_mdct(mrc_handle* h, int64_t n_blocks, int a, int b, const double* blocks, int apply_window,
             double* lines, int32_t* overall_scale);
/* psychoac.py:176-219 (CalcSMRs): smr[n_blocks][nBands]; thresh (masked threshold,
 * psychoac.py:134-173, [n_blocks][N/2]) may be NULL.  scaled_lines [n_blocks][N/2] are the MDCT lines
 * already multiplied by 2^overall_scale[i] as CalcSMRs receives them; pass scaled_lines = NULL (and
 * overall_scale = NULL) to have them computed from `blocks` by the windowed MDCT. */
int mrc_smr(mrc_handle* h, int64_t n_blocks, int a, int b, const double* blocks,
            const double* scaled_lines, const int32_t* overall_scale, double* smr, double* thresh);
/* bitalloc.py:106-155 for n_cases independent problems of n_bands (<= 64) bands each:
 * smr [n_cases][n_bands] (not modified), n_lines [n_bands], budget [n_cases].
 * bits [n_cases][n_bands], bits_left [n_cases] (int(bitsLeft), truncated toward zero). */
int mrc_bitalloc(mrc_handle* h, int64_t n_cases, int n_bands, int max_mant_bits, const int32_t* n_lines,
                 const double* budget, const double* smr, int32_t* bits, int32_t* bits_left);
/* The same with bitalloc.py:132-151's SIDE EFFECT: smr [n_cases][n_bands] is overwritten with the running values the
 * loop leaves behind (-12 for a band's first grant, -6 per further bit, -99999999999999999.0 once retired). */
int mrc_bitalloc_inplace(mrc_handle* h, int64_t n_cases, int n_bands, int max_mant_bits, const int32_t* n_lines,
                         const double* budget, double* smr, int32_t* bits, int32_t* bits_left);
/* quantize.py:114-146 elementwise: scale[i] = ScaleFactor(v[i], n_scale_bits, n_mant_bits[i]). */
int mrc_scale_factor(mrc_handle* h, int64_t n, int n_scale_bits, const double* v, const int32_t* n_mant_bits,
                     int32_t* scale);
/* quantize.py:294-322 elementwise: mant[i] = vMantissa([x[i]], scale[i], n_scale_bits, n_mant_bits[i]). */
int mrc_mantissa(mrc_handle* h, int64_t n, int n_scale_bits, const double* x, const int32_t* scale,
                 const int32_t* n_mant_bits, int32_t* mant);
/* quantize.py:61-87 (vQuantizeUniform; QuantizeUniform 12-38 is its scalar form) elementwise: code[i] = sign bit <<
 * (n_bits - 1) + magnitude code of |x[i]|; 1 <= n_bits <= 62. */
int mrc_quantize_uniform(mrc_handle* h, int64_t n, int n_bits, const double* x, int64_t* code);
/* psychoac.py:27-29 elementwise: z = 13 atan(0.76 f / 1000) + 3.5 atan((f / 7500)^2). */
int mrc_bark(mrc_handle* h, int64_t n, const double* f, double* z);
/* pcmfile.py:91-100 elementwise: out[i] = sign(c) 2|c| / 65535 (one rounding), -32768 -> 0.0 -- the map the PCM16
 * ingest paths apply on load. */
int mrc_pcm_to_float(mrc_handle* h, int64_t n, const int16_t* pcm, double* out);
/* pacfileThem.py:1025-1056 (TransientDetector), numeric part, for every hop of a stream at once: streams is
 * [n_channels][(n_hops+1)*nMDCTLines] (each channel starts with the prior hop); hop i = samples
 * [(i+1)*nMDCTLines, (i+2)*nMDCTLines) is filtered FROM A ZERO STATE (the reference calls sosfilt without zi)
 * by the n_sections second-order sections sos[n_sections][6] = {b0,b1,b2,a0=1,a1,a2} (scipy layout; the
 * reference designs them with signal.cheby2(20,40,9000/fs,'high') + tf2sos, pacfileThem.py:1146-1147).
 * peaks [n_hops][n_channels][nMDCTLines/nSamplesShort + 1]: max |y| of each short sub-block, then of the hop.
 * The threshold tests and the block-shape sequencing (pacfileThem.py:1046-1056, 1182-1214) are host logic:
 * mrcaudiocodec_amd/transient.py. */
int mrc_transient_peaks(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                        const double* streams, double* peaks);
/* ... with the streams given as MRC_SAMPLES_F64 (double*) or MRC_SAMPLES_PCM16 (int16_t*: the WAV's own codes, converted
 * on load as pcmfile.py:91-100 does) */
int mrc_transient_peaks_ex(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                           const void* streams, int sample_format, double* peaks);
/* The same on streams that are already in DEVICE memory, as float64 signed fractions or as the file's int16 PCM codes
 * (converted on load, pcmfile.py:91-100): channel c starts at streams + c * channel_stride samples; peaks (device)
 * [n_hops][n_channels][nMDCTLines/nSamplesShort + 1]; sos is a HOST pointer.  Enqueued on `stream`.  The filter
 * coefficients are staged in a buffer the HANDLE owns: one stream per handle at a time -- a second call on another stream
 * while the first may still run would overwrite (or, growing, free) what its kernel reads; use one handle per stream. */
int mrc_dev_transient_peaks(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                            const void* streams, int sample_format, int64_t channel_stride, double* peaks, void* stream);
/* ms_stereo.py:53-67 elementwise over n lines: out_mid = max(mid, min(side, MLD side)), out_side likewise, MLD from z
 * (Bark).  Kept for the drop-in module's symbol; the encoder's own use of it is dead (psychoac.py:205-210). */
int mrc_stereo_masking_factor(mrc_handle* h, int64_t n, const double* mid_thresh, const double* side_thresh,
                              const double* z, double* out_mid, double* out_side);
/* ms_stereo.py:5-27 for n_blocks pairs of line vectors [n_blocks][n_total_lines] and one band table. */
int mrc_ms_switch(mrc_handle* h, int64_t n_blocks, int n_bands, const int32_t* n_lines,
                  const double* lines_left, const double* lines_right, int32_t* ms_switch);

/* ---- device entry points (batch / stream mode; what bench.py and multi-GPU sharding drive) ------ */

/* Frame f of the batch reads its a+b samples at ch[offsets ? offsets[f] : f*frame_stride ...].
 * frame_stride = b  -> an overlapped PCM stream, every hop read once (pacfileThem.py:628-631);
 * frame_stride = a+b -> explicit blocks.  ch_right == NULL -> mono (nsig = 1), else joint (nsig = 4).
 * Buffers (device): lines [n][nsig][N/2] f64, overall_scale [n][nsig] i32, smr [n][nsig][nBands] f64,
 * ms_switch [n][nBands] i32 (joint only), bit_alloc/scale_factor [n][nstream][nBands] i32,
 * mantissa [n][nstream][N/2] i32, reservoir_in (may be NULL) / reservoir_out [n] i32. */
int mrc_dev_mdct(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                 int64_t frame_stride, const int64_t* offsets, double* lines, int32_t* overall_scale, void* stream);
int mrc_dev_smr(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                int64_t frame_stride, const int64_t* offsets, const double* lines, const int32_t* overall_scale,
                double* smr, double* thresh /*nullable*/, void* stream);
int mrc_dev_alloc_quant(mrc_handle* h, int a, int b, int64_t n_frames, int joint,
                        const double* lines, const int32_t* overall_scale, const double* smr,
                        const int32_t* reservoir_in, int32_t* ms_switch, int32_t* bit_alloc,
                        int32_t* scale_factor, int32_t* mantissa, int32_t* reservoir_out, void* stream);
/* The three stages back to back, intermediates in the handle's workspace (grown on demand, never
 * inside a timed loop once warm).  lines_out may be NULL. */
int mrc_dev_encode(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                   int64_t frame_stride, const int64_t* offsets, const int32_t* reservoir_in,
                   int32_t* overall_scale, int32_t* ms_switch, int32_t* bit_alloc, int32_t* scale_factor,
                   int32_t* mantissa, int32_t* reservoir_out, double* lines_out, void* stream);
/* The same with the channel format (MRC_SAMPLES_*: ch_left / ch_right are double* or int16_t*) and the mantissa
 * plane format (MRC_MANTISSA_*: int32_t* or uint16_t*) chosen by the caller; offsets / frame_stride count samples. */
int mrc_dev_encode_ex(mrc_handle* h, int a, int b, int64_t n_frames, const void* ch_left, const void* ch_right,
                      int sample_format, int64_t frame_stride, const int64_t* offsets, const int32_t* reservoir_in,
                      int32_t* overall_scale, int32_t* ms_switch, int32_t* bit_alloc, int32_t* scale_factor,
                      void* mantissa, int mantissa_format, int32_t* reservoir_out, double* lines_out, void* stream);

/* ---- host-side back end: Huffman table choice + `.pac` bit packing (no GPU, no handle) -----------
 * BASELINE.json's north_star keeps huffman.py / bitpack.py on the host; these are their C++ form, fed with
 * the dense outputs of mrc_encode_mono / mrc_encode_joint.  Table ids: sorted names (percussive 0,
 * silence 1, speech 2, tonal 3), 15 = raw mantissas (codecThem.py:149). */

/* psychoac.py:86-105 + pacfileThem.py:637-645: lines per scale-factor band of block shape (a,b). */
int mrc_band_table(const mrc_config* cfg, int a, int b, int32_t* n_bands, int32_t* n_lines /*[MRC_MAX_BANDS]*/);
/* Upper bound of the bytes one block can pack to (all its channel chunks, length prefixes included). */
int64_t mrc_pack_bound(const mrc_config* cfg, int a, int b, int n_channels, int joint);
/* pacfileThem.py:586-613 (file header; num_samples is padded by the reference's inverted test). */
int mrc_pac_header(const mrc_config* cfg, int n_channels, uint32_t num_samples, uint8_t* out, int64_t out_cap,
                   int64_t* out_len);
/* Host threads used by mrc_pack_* / mrc_unpack_blocks (process-wide).  Default: the CPUs this process may run on
 * (sched_getaffinity), at most 16, or $MRC_PACK_THREADS. */
int mrc_pack_set_threads(int n_threads);
int mrc_pack_get_threads(void);
/* What PACFile.WriteDataBlock appends per block (pacfileThem.py:652-790) for n_channels independent
 * channels: per channel `<L nBytes` + MSB-first payload {huffTable:4, blkswA, blkswB, overallScale, band
 * records}.  Arrays: overall_scale [n][nch], scale_factor / bit_alloc [n][nch][nBands], mantissa
 * [n][nch][N/2] dense.  use_huffman = 0 writes raw mantissas (EncodeNoHuff), else the table choice of
 * codecThem.py:136-203 runs per channel; huff_table / bits_saved ([n][nch], may be NULL) return the chosen
 * table id and the reservoir credit `bits_saved` (codecThem.py:202,224).  block_offset [n+1] = byte
 * offsets of the blocks in `out`.  Returns MRC_ERR_NOMEM if out_cap is too small. */
int mrc_pack_blocks(const mrc_config* cfg, int64_t n_blocks, int n_channels, int a, int b, int use_huffman,
                    const int32_t* overall_scale, const int32_t* scale_factor, const int32_t* bit_alloc,
                    const int32_t* mantissa, uint8_t* out, int64_t out_cap, int64_t* block_offset,
                    int32_t* huff_table, int32_t* bits_saved);
/* What PACFile.JointWriteDataBlock appends per block (pacfileThem.py:825-970): channel 0 additionally
 * carries overallScale[L,R,M,S] and ms_switch.  overall_scale [n][4], ms_switch [n][nBands], the rest
 * [n][2][...]. */
int mrc_pack_joint_blocks(const mrc_config* cfg, int64_t n_blocks, int a, int b, int use_huffman,
                          const int32_t* overall_scale, const int32_t* ms_switch, const int32_t* scale_factor,
                          const int32_t* bit_alloc, const int32_t* mantissa, uint8_t* out, int64_t out_cap,
                          int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved);

/* The same bytes with the table of every channel chunk GIVEN (huff_table_in [n][nch], ids 0..3 or 15 = raw) -- e.g.
 * the choice mrc_dev_huffman_gain already made on the device -- so the host does not price the four tables again
 * (codecThem.py:151-179); only the recoding and bit packing of codecThem.py:182-200 / pacfileThem.py:716-781,
 * 892-963 remain.  MRC_ERR_INVALID for an id outside {0..3, 15}. */
int mrc_pack_blocks_with_tables(const mrc_config* cfg, int64_t n_blocks, int n_channels, int a, int b,
                                const int32_t* huff_table_in, const int32_t* overall_scale, const int32_t* scale_factor,
                                const int32_t* bit_alloc, const int32_t* mantissa, uint8_t* out, int64_t out_cap,
                                int64_t* block_offset);
int mrc_pack_joint_blocks_with_tables(const mrc_config* cfg, int64_t n_blocks, int a, int b, const int32_t* huff_table_in,
                                      const int32_t* overall_scale, const int32_t* ms_switch,
                                      const int32_t* scale_factor, const int32_t* bit_alloc, const int32_t* mantissa,
                                      uint8_t* out, int64_t out_cap, int64_t* block_offset);

/* The general form of the four packers above: joint = 0 / 1, huff_table_in NULL (price on the host when use_huffman)
 * or given, and the mantissa plane in either format the encoder writes (MRC_MANTISSA_I32: int32_t*, MRC_MANTISSA_I16:
 * uint16_t*, what mrc_encode_stream_pcm16 and mrc_dev_encode_ex deliver) -- no widening copy in between. */
int mrc_pack_blocks_ex(const mrc_config* cfg, int64_t n_blocks, int n_channels, int a, int b, int joint, int use_huffman,
                       const int32_t* huff_table_in, const int32_t* overall_scale, const int32_t* ms_switch,
                       const int32_t* scale_factor, const int32_t* bit_alloc, const void* mantissa, int mantissa_format,
                       uint8_t* out, int64_t out_cap, int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved);

/* Huffman table PRICING on the device (codecThem.py:136-180,202): per (frame, stream) the id of the cheapest
 * table (15 = raw) and bits_saved.  Lets a chained multi-stream encode carry the reservoir from block to block
 * without leaving the GPU; the bytes are produced later by mrc_pack_*.  bit_alloc [n][n_streams][nBands],
 * mantissa [n][n_streams][N/2] dense.  Device form: reservoir_next[f] (may be NULL) = reservoir_out[f] + sum
 * over the frame's streams of bits_saved -- the reservoir the stream's next block starts from (codecThem.py:274). */
int mrc_huffman_gain(mrc_handle* h, int64_t n_blocks, int a, int b, int n_streams, const int32_t* bit_alloc,
                     const int32_t* mantissa, int32_t* huff_table, int32_t* bits_saved);
int mrc_dev_huffman_gain(mrc_handle* h, int a, int b, int64_t n_frames, int n_streams, const int32_t* bit_alloc,
                         const int32_t* mantissa, const int32_t* reservoir_out, int32_t* huff_table,
                         int32_t* bits_saved, int32_t* reservoir_next, void* stream);
/* The `.pac` chunks of n_blocks blocks of ONE shape, packed ON THE DEVICE: the bytes WriteDataBlock /
 * JointWriteDataBlock append per block (pacfileThem.py:652-781, 825-963; bitpack.py:36-101; the Huffman choice
 * and recoding of codecThem.py:136-203) -- byte for byte what mrc_pack_blocks_ex writes on the host, from the
 * encoder's outputs where mrc_dev_encode* left them.  All array arguments are DEVICE pointers, layouts as
 * mrc_pack_blocks_ex: overall_scale [n][joint ? 4 : n_channels], ms_switch [n][nBands] (joint), scale_factor /
 * bit_alloc [n][n_channels][nBands], mantissa [n][n_channels][(a+b)/2] (MRC_MANTISSA_I32 / _I16), huff_table_in
 * [n * n_channels] or NULL (NULL: priced here if use_huffman, else raw).  out [out_cap] receives the chunks
 * (4-byte length + payload each), block_offset [n + 1] the byte offset of every block (last entry: the total);
 * huff_table / bits_saved [n * n_channels] may be NULL.  total_bytes (HOST pointer) non-NULL: the call waits for
 * the stream and returns the total, MRC_ERR_NOMEM if it exceeds out_cap (nothing is written past out_cap; size
 * the buffer with mrc_pack_bound), MRC_ERR_INVALID for a table id outside {0..3, 15} or a chunk beyond mrc_pack_bound
 * (bit_alloc > 16 handed in); NULL: fully asynchronous, read block_offset[n] later -- the packer's workspace belongs to
 * the handle, so asynchronous calls on one handle must all be queued on ONE stream, and errors of an asynchronous
 * call are only seen by mrc_dev_pack_status. */
int mrc_dev_pack_blocks(mrc_handle* h, int a, int b, int64_t n_blocks, int n_channels, int joint, int use_huffman,
                        const int32_t* huff_table_in, const int32_t* overall_scale, const int32_t* ms_switch,
                        const int32_t* scale_factor, const int32_t* bit_alloc, const void* mantissa, int mantissa_format,
                        uint8_t* out, int64_t out_cap, int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved,
                        int64_t* total_bytes, void* stream);
/* Waits for `stream` and returns the status of the most recent mrc_dev_pack_blocks on the handle: MRC_OK,
 * MRC_ERR_INVALID (bad table id / chunk beyond the bound) or MRC_ERR_NOMEM (out_cap exceeded); total_bytes (nullable). */
int mrc_dev_pack_status(mrc_handle* h, int64_t* total_bytes, void* stream);

/* ---- chained stream encode: the reference's WHOLE encode loop in one call ---------------------------------------
 * What `python pacfileThem.py in.wav` does in its encode direction for n_streams stereo streams at once -- one
 * JointWriteDataBlock per block with codingParams.bitReservoir carried from block to block (pacfileThem.py:1159-1214,
 * 793-972; codecThem.py:262-278, 381-396, 503), then Close()'s block through the non-joint writer (973-984), behind the
 * file header (586-613) -- given the block shapes (the transient detector's decisions: mrc_transient_peaks +
 * mrcaudiocodec_amd/transient.py).  The reference runs the whole kernel set once per block because block t+1's bit budget
 * contains what block t left over; here everything that does not depend on the reservoir (window, MDCT, overall
 * scale, SMRs, M/S switch: 95 % of the work) runs as ONE batch per block shape over all blocks of all streams, and the
 * rest (bit allocation -> scale factors -> mantissas -> Huffman pricing -> next reservoir) is a serial scan per stream
 * ON THE DEVICE, one workgroup per stream, no host round trip and no launch per block; the device packer then writes
 * the chunks in file order.
 *   pcm_left / pcm_right [n_streams][stream_stride] int16 PCM codes (converted on load as pcmfile.py:91-100 does); each
 *     stream starts with its prior hop (zeros at file start, pacfileThem.py:615-618).
 *   Blocks of stream s: indices block_start[s] .. block_start[s + 1] - 1 of block_offset / block_a / block_b: block i
 *     reads block_a[i] + block_b[i] samples at block_offset[i] of its stream (the a samples carried over, then the b new
 *     ones, pacfileThem.py:799-802); shapes are the four of the reference's block switching, (L, L), (L, S), (S, S),
 *     (S, L) with L = n_mdct_lines, S = n_short.
 *   reservoir_in [n_streams] (NULL: zeros): codingParams.bitReservoir at the start (pacfileThem.py:1111), e.g. handed
 *     over from the previous shard of a long stream.
 *   use_huffman = 0: EncodeNoHuff's raw mantissas (table id 15).  with_flush: append Close()'s two non-joint chunks
 *     per stream (the stream must then end with a long block, as the reference's Close() assumes).
 *   num_samples [n_streams] (NULL: no headers): the header's sample count (the WAV's, pacfileThem.py:1103); the header
 *     of stream s is written in front of its first chunk, so out[stream_byte_offset[s] .. stream_byte_offset[s + 1])
 *     is the complete `.pac` file of stream s.
 *   out [out_cap]: size it with mrc_chain_out_bound, or less and retry on MRC_ERR_NOMEM (total_bytes then holds the
 *     size needed).  item_byte_offset (NULL or [n_items + 1], n_items = blocks + 2 n_streams with_flush): where every
 *     block's chunks start (asked for, the position of every chunk is read back from the device: 8 bytes per chunk; with
 *     NULL only the stream starts come back).  reservoir_out (NULL or [n_streams]): codingParams.bitReservoir after the
 *     last block.
 *     reservoir_trace (NULL or [n_items]): ... after every block (tests).
 * mrc_dev_encode_chained_pac: the same with pcm_left / pcm_right / out in DEVICE memory (all other pointers host);
 * it synchronises `stream` before it returns (the byte offsets come back).  mrc_get_chain_ms: device time of the last
 * chained call -- ms[0] phase A + preparation, ms[1] the serial scan, ms[2] packing, ms[3] all three. */
int64_t mrc_chain_out_bound(mrc_handle* h, int64_t n_streams, const int64_t* block_start, const int32_t* block_a,
                            const int32_t* block_b, int with_flush, int with_headers);
int mrc_encode_chained_stream_pcm16_pac(mrc_handle* h, int64_t n_streams, const int16_t* pcm_left, const int16_t* pcm_right,
                                        int64_t stream_stride, const int64_t* block_start, const int64_t* block_offset,
                                        const int32_t* block_a, const int32_t* block_b, const int32_t* reservoir_in,
                                        int use_huffman, int with_flush, const uint32_t* num_samples, uint8_t* out,
                                        int64_t out_cap, int64_t* stream_byte_offset, int64_t* item_byte_offset,
                                        int32_t* reservoir_out, int32_t* reservoir_trace, int64_t* total_bytes);
/* ... with the samples as MRC_SAMPLES_PCM16 (int16_t*) or MRC_SAMPLES_F64 (double*: the signed fractions pcmfile.py:98
 * hands the codec -- what the per-block API takes) */
int mrc_encode_chained_stream_pac(mrc_handle* h, int64_t n_streams, const void* pcm_left, const void* pcm_right,
                                  int sample_format, int64_t stream_stride, const int64_t* block_start,
                                  const int64_t* block_offset, const int32_t* block_a, const int32_t* block_b,
                                  const int32_t* reservoir_in, int use_huffman, int with_flush, const uint32_t* num_samples,
                                  uint8_t* out, int64_t out_cap, int64_t* stream_byte_offset, int64_t* item_byte_offset,
                                  int32_t* reservoir_out, int32_t* reservoir_trace, int64_t* total_bytes);
int mrc_dev_encode_chained_pac(mrc_handle* h, int64_t n_streams, const void* pcm_left, const void* pcm_right,
                               int sample_format, int64_t stream_stride, const int64_t* block_start,
                               const int64_t* block_offset, const int32_t* block_a, const int32_t* block_b,
                               const int32_t* reservoir_in, int use_huffman, int with_flush, const uint32_t* num_samples,
                               uint8_t* out, int64_t out_cap, int64_t* stream_byte_offset, int64_t* item_byte_offset,
                               int32_t* reservoir_out, int32_t* reservoir_trace, int64_t* total_bytes, void* stream);
int mrc_get_chain_ms(mrc_handle* h, double* ms /*[4]*/);

/* ---- sensitivity certificate (round 4) ----
 * Bit-identity of the integers with the reference is an empirical, counted result: each of them is a floor / compare of
 * float64 values whose last bits differ between implementations (FFT factorisation, log10 / atan / 2^x), and it can only come
 * out differently where the deciding value lies within that difference of its edge.  With MRC_OPT_SENSITIVITY set, the
 * encode calls of a handle count those places; counts [MRC_SENS_COUNT] accumulate until reset:
 *   MRC_SENS_QUANT     mantissa codes within 4e-13 (of the block's scaled peak) of their truncation edge, and scale factors
 *                      decided by a band peak that close to a power-of-two code (quantize.py:12-38,114-146,294-322)
 *   MRC_SENS_BITALLOC  pairs of bands whose SMRs are a multiple of 6 dB apart to within 1e-9 dB: their running values tie
 *                      at some step of the greedy loop (bitalloc.py:132-151, np.argmax)
 *   MRC_SENS_MS        bands whose M/S test is within 1e-12 (relative) of its 0.8 threshold (ms_stereo.py:5-27)
 *   MRC_SENS_PEAK      spectral bins within 1e-11 (relative) of a neighbour they have to beat strictly (psychoac.py:162)
 *   MRC_SENS_NODES     64-line chunks the slope-node evaluation of the masking sum sent back to the sorted sweep (its error
 *                      bound exceeded 1e-13 of a line's masked intensity) -- informational: the result is then the sweep's
 *   MRC_SENS_FRAMES    blocks examined
 * A call whose first four counts are zero took no decision near an edge.  Not counted: the overall scale (a 20-bit code of
 * the block peak), the transient detector's threshold tests.  Synchronises the handle's stream. */
#define MRC_SENS_QUANT 0
#define MRC_SENS_BITALLOC 1
#define MRC_SENS_MS 2
#define MRC_SENS_PEAK 3
#define MRC_SENS_NODES 4
#define MRC_SENS_FRAMES 5
#define MRC_SENS_COUNT 8
int mrc_get_sensitivity(mrc_handle* h, int64_t* counts /*[MRC_SENS_COUNT]*/, int reset);
/* mrc_chain_fetch_output: the bytes of the LAST mrc_encode_chained_stream[_pcm16]_pac call on this handle, which stay in the
 * handle's device buffer until the next chained call: after MRC_ERR_NOMEM ("out_cap too small") a caller allocates
 * total_bytes and fetches them -- no second encode.  The offsets / reservoirs of that call were already returned by it. */
int mrc_chain_fetch_output(mrc_handle* h, uint8_t* out, int64_t out_cap, int64_t* total_bytes);

/* ---- decode side ("next" row f-4: the reference's decoder, pacfileThem.py:130-585 + codecThem.py:30-134) ----
 * Host: header and chunk parsing (no GPU, no handle).  Device: dequantise -> undo the overall scale -> M/S
 * reconstruction -> IMDCT -> transition window -> overlap-and-add, and the 16-bit PCM codes. */

/* pacfileThem.py:130-158.  Fills sample_rate, n_mdct_lines, n_scale_bits, n_mant_size_bits of *cfg (other fields
 * untouched); data_offset = first chunk. */
int mrc_pac_read_header(const uint8_t* buf, int64_t len, mrc_config* cfg, int32_t* n_channels, uint32_t* num_samples,
                        int64_t* data_offset);
/* Offsets of the `<L nBytes` + payload chunks after the header; returns their number (chunk_offset may be NULL /
 * cap 0 to count), < 0 if a chunk is truncated. */
int64_t mrc_pac_scan_chunks(const uint8_t* buf, int64_t len, int64_t data_offset, int64_t* chunk_offset, int64_t cap);
/* The parsing half of PACFile.ReadDataBlock (pacfileThem.py:176-302, joint = 0) / JointReadDataBlock (341-560,
 * joint = 1, two chunks per block): table id, block-switch bits -> (a, b), overall scale(s), M/S switch, band
 * records with raw or Huffman-coded mantissas (prefix decoding by table; ids in sorted-name order).  Fixed strides
 * because the shape varies from block to block: overall_scale [n][joint ? 4 : n_channels], ms_switch
 * [n][MRC_MAX_BANDS], huff_table [n][n_channels], scale_factor / bit_alloc [n][n_channels][MRC_MAX_BANDS], mantissa
 * [n][n_channels][n_mdct_lines] dense.  chunk_offset [n * n_channels]. */
int mrc_unpack_blocks(const mrc_config* cfg, int64_t n_blocks, int n_channels, int joint, const uint8_t* buf, int64_t len,
                      const int64_t* chunk_offset, int32_t* a, int32_t* b, int32_t* huff_table, int32_t* overall_scale,
                      int32_t* ms_switch, int32_t* scale_factor, int32_t* bit_alloc, int32_t* mantissa);

/* codecThem.Decode (30-63; n_streams = 1, overall_scale [n]) / JointDecode (65-134; n_streams = 2, overall_scale
 * [n][4] = L,R,M,S, ms_switch [n][nBands]) for n blocks of shape (a, b): scale_factor / bit_alloc [n][n_streams][nBands],
 * mantissa [n][n_streams][N/2] dense -> windowed blocks out [n][n_streams][a+b] (before overlap-and-add). */
int mrc_decode(mrc_handle* h, int64_t n_blocks, int a, int b, int n_streams, const int32_t* overall_scale,
               const int32_t* ms_switch, const int32_t* scale_factor, const int32_t* bit_alloc, const int32_t* mantissa,
               double* out);
/* Device form with the overlap-and-add of pacfileThem.py:312-315 fused: block i ADDS its a+b windowed samples to
 * out_left (and out_right) starting at sample out_offset[i]; the caller zeroes the outputs first.  Every sample
 * receives at most two contributions, so the sum does not depend on the order of the blocks. */
int mrc_dev_decode(mrc_handle* h, int a, int b, int64_t n_blocks, int n_streams, const int32_t* overall_scale,
                   const int32_t* ms_switch, const int32_t* scale_factor, const int32_t* bit_alloc,
                   const int32_t* mantissa, const int64_t* out_offset, double* out_left, double* out_right, void* stream);
/* pcmfile.py:163-172: signed fractions -> 16-bit PCM codes (sign-magnitude quantiser of quantize.py:61-87, then
 * 2's complement).  mrc_pcm16: host pointers; mrc_dev_pcm16: device pointers. */
int mrc_pcm16(mrc_handle* h, int64_t n, const double* x, int16_t* out);
int mrc_dev_pcm16(mrc_handle* h, int64_t n, const double* x, int16_t* out, void* stream);

/* Per-stage device time of the most recent mrc_dev_encode / stage call when timing is enabled
 * (hipEvents on the launch stream; the call then synchronises).  ms[0..2] = mdct, smr, alloc+quant. */
int mrc_set_timing(mrc_handle* h, int enabled);
/* Options.  MRC_OPT_EXACT_SPREAD = 1: evaluate the masker spreading (psychoac.py:68-78) with the
 * reference's own expression and pow() per (masker, line) instead of the factored fast form (default 0).
 * Both give the same integer outputs on the parity corpora; the exact form is ~8x slower. */
#define MRC_OPT_EXACT_SPREAD 1
/* MRC_OPT_SMR_ALL_BANDS = 1: in joint blocks compute the SMRs of all four signals in every band (default 0: only the
 * pair the M/S switch selects per band, ms_stereo.py:70-81 -- the other pair never reaches the bit allocation). */
#define MRC_OPT_SMR_ALL_BANDS 2
/* MRC_OPT_CHAIN_FORCE_REPAIR = 1 (tests only): the chained encode's event preparation scrambles its candidate order so
 * that the repair pass, which otherwise only runs on near-ties that rounding turned round, does real work. */
#define MRC_OPT_CHAIN_FORCE_REPAIR 3
/* MRC_OPT_CHAIN_THREADS = 0 | 256 | 512 | 1024: threads of the workgroup that walks one stream in the chained encode's
 * serial scan (default 0: 512 for up to 512 streams -- the latency of the one stream counts -- else 256: eight streams per CU). */
#define MRC_OPT_CHAIN_THREADS 4
/* MRC_OPT_SENSITIVITY = 1: every encode call on the handle also counts the integer decisions it took within a guard band of
 * floating-point rounding (see mrc_get_sensitivity); costs a pass over the intermediate results (~10 % of an encode).
 * = 2 (tests): the same with every guard band 10^8 times wider, so that an ordinary corpus produces counts. */
#define MRC_OPT_SENSITIVITY 5
/* MRC_OPT_CHAIN_SLAB_BLOCKS = n: the chained encode (mrc_encode_chained_stream*_pac, mrc_dev_encode_chained_pac) cuts a call
 * into slabs of at most n blocks -- whole streams while they fit, a longer stream alone in consecutive time slabs, the bit
 * reservoir carried from slab to slab -- so that the device memory of a call is bounded by the slab (~45 KB per joint long
 * block + the slab's worst-case output, 13 KB per block) whatever the length of the file.  Default 131 072 (~7.5 GB; a
 * slab costs ~0.25 ms of host work between its neighbours: 8 192 streams x 12 blocks run 5 % slower in two slabs than in
 * one); 65 536: < 4 GB; 0: one slab.  The bytes do not depend on it. */
#define MRC_OPT_CHAIN_SLAB_BLOCKS 6
int mrc_set_option(mrc_handle* h, int option, int value);
int mrc_get_option(mrc_handle* h, int option, int32_t* value);
int mrc_get_stage_ms(mrc_handle* h, double* ms /*[3]*/);
/* ... and per kernel: ms[0..4] = MDCT, smr_kernel, band_stats_kernel (joint only, else ~0), bitalloc_kernel,
 * quantize_kernel. */
int mrc_get_kernel_ms(mrc_handle* h, double* ms /*[5]*/);

#ifdef __cplusplus
}
#endif
#endif /* MRC_HIP_H */
