// C ABI of libmrc_hip.so (include/mrc_hip.h): handle, shape cache, workspace, host<->device staging.
// No computation happens here and there is no CPU fallback: every entry point ends in a kernel launch
// of mrc_kernels.hip or fails.
#include "mrc_handle.hpp"

#include <cstdio>
#include <cstring>
#include <new>

using namespace mrc;

namespace mrc {

std::string& create_error() {
    static std::string s;
    return s;
}

int get_shape(mrc_handle* h, int a, int b, const HostShape** out) {
    MRC_HIP(h, hipSetDevice(h->device));
    auto key = std::make_pair(a, b);
    auto it = h->shapes.find(key);
    if (it == h->shapes.end()) {
        HostShape hs;
        std::string err;
        if (!build_shape(h->cfg, a, b, &hs, &err)) return fail(h, MRC_ERR_INVALID, err);
        it = h->shapes.emplace(key, std::move(hs)).first;
    }
    *out = &it->second;
    return MRC_OK;
}

}  // namespace mrc

extern "C" {

int mrc_version(void) { return MRC_VERSION; }

int mrc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mrc_default_config(mrc_config* cfg) {
    if (!cfg) return;
    cfg->sample_rate = 48000;
    cfg->n_mdct_lines = 1024;          // pacfileThem.py:1105
    cfg->n_short = 128;                // pacfileThem.py:1114
    cfg->n_scale_bits = 4;             // pacfileThem.py:1106
    cfg->n_mant_size_bits = 4;         // pacfileThem.py:1107
    cfg->blksw_bits_a = 1;             // pacfileThem.py:1119-1120
    cfg->blksw_bits_b = 1;
    cfg->device_id = 0;
    cfg->target_bits_per_sample = 2.86;  // pacfileThem.py:1108
}

int mrc_create(const mrc_config* cfg, mrc_handle** out) {
    if (!cfg || !out) return fail(nullptr, MRC_ERR_INVALID, "mrc_create: null argument");
    *out = nullptr;
    if (cfg->sample_rate <= 0 || cfg->n_mdct_lines <= 0 || cfg->n_short <= 0 || cfg->n_scale_bits < 1 ||
        cfg->n_scale_bits > 4 || cfg->n_mant_size_bits < 1 || cfg->n_mant_size_bits > 8)
        return fail(nullptr, MRC_ERR_INVALID, "mrc_create: parameter out of range (nScaleBits in 1..4, nMantSizeBits in 1..8)");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, MRC_ERR_NO_DEVICE, "mrc_create: no HIP device available (this library has no CPU path)");
    if (cfg->device_id < 0 || cfg->device_id >= n) return fail(nullptr, MRC_ERR_INVALID, "mrc_create: bad device_id");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, cfg->device_id)) != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MRC_ERR_NO_DEVICE, std::string("mrc_create: device is ") + prop.gcnArchName +
                                                    ", the kernels are built for gfx950 (MI355X) only");
    mrc_handle* h = new (std::nothrow) mrc_handle();
    if (!h) return fail(nullptr, MRC_ERR_NOMEM, "mrc_create: out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device_id;
    if ((e = hipSetDevice(h->device)) != hipSuccess || (e = hipStreamCreate(&h->stream)) != hipSuccess) {
        delete h;
        return hip_fail(nullptr, e, "hipSetDevice/hipStreamCreate");
    }
    for (auto& ev : h->ev) (void)hipEventCreate(&ev);
    // the four shapes of the reference's block switching (pacfileThem.py:1192-1210)
    const int L = cfg->n_mdct_lines, Sh = cfg->n_short;
    const int shapes[4][2] = {{L, L}, {Sh, Sh}, {L, Sh}, {Sh, L}};
    for (auto& s : shapes) {
        const HostShape* hs;
        int rc = get_shape(h, s[0], s[1], &hs);
        if (rc != MRC_OK) { create_error() = h->error; mrc_destroy(h); return rc; }
    }
    *out = h;
    return MRC_OK;
}

void mrc_destroy(mrc_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& kv : h->shapes) free_shape(&kv.second);
    for (hipStream_t* st : {&h->stIn, &h->stOut}) {
        if (*st) { (void)hipStreamSynchronize(*st); (void)hipStreamDestroy(*st); }
        *st = nullptr;
    }
    for (auto& lane : h->lanes) lane.release();
    h->wsPipe.release();
    h->packWs.release();
    h->chain.release();
    h->ws.release();
    for (DevBuf* b : {&h->inL, &h->inR, &h->inAux, &h->inAux2, &h->inAux3,
                      &h->outA, &h->outB, &h->outC, &h->outD, &h->outE, &h->outF, &h->outG})
        b->release();
    h->sens.release();
    h->pinIn.release();
    h->pinOut.release();
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* mrc_last_error(const mrc_handle* h) { return h ? h->error.c_str() : create_error().c_str(); }

int mrc_shape_bands(mrc_handle* h, int a, int b, int32_t* n_bands, int32_t* n_lines) {
    if (!h || !n_bands) return fail(h, MRC_ERR_INVALID, "mrc_shape_bands: null argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    *n_bands = hs->dev.nBands;
    if (n_lines) for (int i = 0; i < hs->dev.nBands; ++i) n_lines[i] = hs->bandN[i];
    return MRC_OK;
}

int mrc_shape_budget(mrc_handle* h, int a, int b, int joint, int32_t reservoir, double* bit_budget) {
    if (!h || !bit_budget) return fail(h, MRC_ERR_INVALID, "mrc_shape_budget: null argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    const DevShape& S = hs->dev;
    double v;
    if (joint) { v = S.budgetJointPre + (double)reservoir; v -= S.blkswA; v -= S.blkswB; }
    else v = S.budgetMono + (double)reservoir;
    *bit_budget = v;
    return MRC_OK;
}

int mrc_set_timing(mrc_handle* h, int enabled) {
    if (!h) return MRC_ERR_INVALID;
    h->timing = enabled != 0;
    return MRC_OK;
}

int mrc_set_option(mrc_handle* h, int option, int value) {
    if (!h) return MRC_ERR_INVALID;
    if (option == MRC_OPT_EXACT_SPREAD) { h->exactSpread = value != 0; return MRC_OK; }
    if (option == MRC_OPT_SMR_ALL_BANDS) { h->smrAllBands = value != 0; return MRC_OK; }
    if (option == MRC_OPT_CHAIN_FORCE_REPAIR) { h->chainForceFallback = value != 0; return MRC_OK; }
    if (option == MRC_OPT_SENSITIVITY) {
        if (value < 0 || value > 2) return fail(h, MRC_ERR_INVALID, "mrc_set_option: MRC_OPT_SENSITIVITY takes 0, 1 or 2");
        if (value) {
            MRC_HIP(h, hipSetDevice(h->device));
            MRC_HIP(h, hipStreamSynchronize(h->stream));
            if (!h->sens.p) {
                MRC_HIP(h, h->sens.reserve(MRC_SENS_COUNT * sizeof(unsigned long long)));
                MRC_HIP(h, hipMemset(h->sens.p, 0, MRC_SENS_COUNT * sizeof(unsigned long long)));
            }
            const double scale = value == 2 ? 1e8 : 1.0;                 // slot 7: the guard scale the kernels read
            MRC_HIP(h, hipMemcpy(h->sens.as<unsigned long long>() + 7, &scale, sizeof scale, hipMemcpyHostToDevice));
        }
        h->sensOn = value != 0;
        return MRC_OK;
    }
    if (option == MRC_OPT_CHAIN_SLAB_BLOCKS) {
        if (value < 0) return fail(h, MRC_ERR_INVALID, "mrc_set_option: MRC_OPT_CHAIN_SLAB_BLOCKS takes a block count (0: no slabs)");
        h->chainSlabBlocks = value;
        return MRC_OK;
    }
    if (option == MRC_OPT_CHAIN_THREADS) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(h, MRC_ERR_INVALID, "mrc_set_option: MRC_OPT_CHAIN_THREADS takes 0, 256, 512 or 1024");
        h->chainThreads = value;
        return MRC_OK;
    }
    return fail(h, MRC_ERR_INVALID, "mrc_set_option: unknown option");
}

int mrc_get_option(mrc_handle* h, int option, int32_t* value) {
    if (!h || !value) return MRC_ERR_INVALID;
    switch (option) {
        case MRC_OPT_EXACT_SPREAD: *value = h->exactSpread ? 1 : 0; return MRC_OK;
        case MRC_OPT_SMR_ALL_BANDS: *value = h->smrAllBands ? 1 : 0; return MRC_OK;
        case MRC_OPT_CHAIN_FORCE_REPAIR: *value = h->chainForceFallback ? 1 : 0; return MRC_OK;
        case MRC_OPT_CHAIN_THREADS: *value = h->chainThreads; return MRC_OK;
        case MRC_OPT_SENSITIVITY: *value = h->sensOn ? 1 : 0; return MRC_OK;
        case MRC_OPT_CHAIN_SLAB_BLOCKS: *value = (int32_t)h->chainSlabBlocks; return MRC_OK;
        default: return fail(h, MRC_ERR_INVALID, "mrc_get_option: unknown option");
    }
}

int mrc_get_sensitivity(mrc_handle* h, int64_t* counts, int reset) {
    if (!h || !counts) return MRC_ERR_INVALID;
    for (int i = 0; i < MRC_SENS_COUNT; ++i) counts[i] = 0;
    if (!h->sens.p) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    unsigned long long v[MRC_SENS_COUNT];
    MRC_HIP(h, hipMemcpy(v, h->sens.p, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < MRC_SENS_COUNT - 1; ++i) counts[i] = (int64_t)v[i];   // (slot 7 is the guard scale)
    if (reset) MRC_HIP(h, hipMemset(h->sens.p, 0, (MRC_SENS_COUNT - 1) * sizeof(unsigned long long)));
    return MRC_OK;
}

int mrc_get_stage_ms(mrc_handle* h, double* ms) {
    if (!h || !ms) return MRC_ERR_INVALID;
    for (int i = 0; i < 3; ++i) ms[i] = h->stageMs[i];
    return MRC_OK;
}

int mrc_get_kernel_ms(mrc_handle* h, double* ms) {
    if (!h || !ms) return MRC_ERR_INVALID;
    for (int i = 0; i < 5; ++i) ms[i] = h->kernelMs[i];
    return MRC_OK;
}

// ---------------------------------------------------------------------------------------------- device API

int mrc_dev_mdct(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                 int64_t frame_stride, const int64_t* offsets, double* lines, int32_t* overall_scale, void* stream) {
    if (!h || !ch_left || !lines || !overall_scale || n_frames < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_dev_mdct: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    MRC_HIP(h, launch_mdct(hs->dev, n_frames, ch_left, ch_right, kSampleF64, frame_stride, offsets, true, lines,
                           overall_scale, pick_stream(h, stream)));
    return MRC_OK;
}

int mrc_dev_smr(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                int64_t frame_stride, const int64_t* offsets, const double* lines, const int32_t* overall_scale,
                double* smr, double* thresh, void* stream) {
    if (!h || !ch_left || !lines || !overall_scale || !smr || n_frames < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_dev_smr: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    MRC_HIP(h, launch_smr(hs->dev, n_frames, ch_left, ch_right, kSampleF64, frame_stride, offsets, lines, overall_scale,
                          smr, thresh, nullptr, nullptr, h->exactSpread, pick_stream(h, stream)));
    return MRC_OK;
}

int mrc_dev_alloc_quant(mrc_handle* h, int a, int b, int64_t n_frames, int joint, const double* lines,
                        const int32_t* overall_scale, const double* smr, const int32_t* reservoir_in,
                        int32_t* ms_switch, int32_t* bit_alloc, int32_t* scale_factor, int32_t* mantissa,
                        int32_t* reservoir_out, void* stream) {
    if (!h || !lines || !overall_scale || !smr || !bit_alloc || !scale_factor || !mantissa || !reservoir_out ||
        (joint && !ms_switch) || n_frames < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_dev_alloc_quant: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    MRC_HIP(h, h->ws.peak.reserve(alloc_workspace_bytes(hs->dev, n_frames, joint)));
    MRC_HIP(h, launch_alloc_quant(hs->dev, n_frames, joint, lines, overall_scale, smr, reservoir_in, ms_switch,
                                  bit_alloc, scale_factor, mantissa, MRC_MANTISSA_I32, reservoir_out,
                                  h->ws.peak.as<double>(), false, false, nullptr, pick_stream(h, stream)));
    return MRC_OK;
}

int mrc_dev_decode(mrc_handle* h, int a, int b, int64_t n_blocks, int n_streams, const int32_t* overall_scale,
                   const int32_t* ms_switch, const int32_t* scale_factor, const int32_t* bit_alloc,
                   const int32_t* mantissa, const int64_t* out_offset, double* out_left, double* out_right, void* stream) {
    if (!h || n_blocks < 0 || (n_streams != 1 && n_streams != 2) || !overall_scale || !scale_factor || !bit_alloc ||
        !mantissa || !out_offset || !out_left || (n_streams == 2 && (!ms_switch || !out_right)))
        return fail(h, MRC_ERR_INVALID, "mrc_dev_decode: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    MRC_HIP(h, launch_decode(hs->dev, n_blocks, n_streams, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa,
                             out_offset, out_left, out_right, pick_stream(h, stream)));
    return MRC_OK;
}

int mrc_dev_pcm16(mrc_handle* h, int64_t n, const double* x, int16_t* out, void* stream) {
    if (!h || n < 0 || !x || !out) return fail(h, MRC_ERR_INVALID, "mrc_dev_pcm16: bad argument");
    MRC_HIP(h, hipSetDevice(h->device));
    MRC_HIP(h, launch_pcm16(n, x, out, pick_stream(h, stream)));
    return MRC_OK;
}

int mrc_dev_huffman_gain(mrc_handle* h, int a, int b, int64_t n_frames, int n_streams, const int32_t* bit_alloc,
                         const int32_t* mantissa, const int32_t* reservoir_out, int32_t* huff_table,
                         int32_t* bits_saved, int32_t* reservoir_next, void* stream) {
    if (!h || !bit_alloc || !mantissa || !huff_table || !bits_saved || n_frames < 0 || n_streams < 1 || n_streams > 4 ||
        (reservoir_next && !reservoir_out))
        return fail(h, MRC_ERR_INVALID, "mrc_dev_huffman_gain: bad argument (1 <= n_streams <= 4)");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    MRC_HIP(h, launch_huffman_gain(hs->dev, n_frames, n_streams, bit_alloc, mantissa, reservoir_out, huff_table,
                                   bits_saved, reservoir_next, pick_stream(h, stream)));
    return MRC_OK;
}

// `.pac` chunks on the device (csrc/mrc_kernels_pack.hip)
int mrc_dev_pack_blocks(mrc_handle* h, int a, int b, int64_t n_blocks, int n_channels, int joint, int use_huffman,
                        const int32_t* huff_table_in, const int32_t* overall_scale, const int32_t* ms_switch,
                        const int32_t* scale_factor, const int32_t* bit_alloc, const void* mantissa, int mantissa_format,
                        uint8_t* out, int64_t out_cap, int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved,
                        int64_t* total_bytes, void* stream) {
    if (!h || n_blocks < 0 || n_channels < 1 || (joint && (n_channels != 2 || !ms_switch)) || !overall_scale ||
        !scale_factor || !bit_alloc || !mantissa || !out || !block_offset || out_cap < 0 ||
        (mantissa_format != MRC_MANTISSA_I32 && mantissa_format != MRC_MANTISSA_I16))
        return fail(h, MRC_ERR_INVALID, "mrc_dev_pack_blocks: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    const DevShape& S = hs->dev;
    const mrc_config& cfg = h->cfg;
    if (cfg.n_scale_bits < 1 || cfg.n_scale_bits > 4 || cfg.n_mant_size_bits < 1 || cfg.n_mant_size_bits > 8 ||
        cfg.blksw_bits_a < 0 || cfg.blksw_bits_a > 8 || cfg.blksw_bits_b < 0 || cfg.blksw_bits_b > 8)
        return fail(h, MRC_ERR_INVALID, "mrc_dev_pack_blocks: field widths out of range");
    hipStream_t st = pick_stream(h, stream);
    const int64_t nChunks = n_blocks * n_channels;
    if (n_blocks == 0) {
        MRC_HIP(h, hipMemsetAsync(block_offset, 0, sizeof(int64_t), st));
        if (total_bytes) { MRC_HIP(h, hipStreamSynchronize(st)); *total_bytes = 0; }
        return MRC_OK;
    }
    static const PackTables tables = [] { PackTables t; pack_tables(&t); return t; }();
    PackParams P;
    P.nch = n_channels; P.joint = joint ? 1 : 0; P.useHuffman = use_huffman ? 1 : 0;
    P.nScaleBits = cfg.n_scale_bits; P.nMantSizeBits = cfg.n_mant_size_bits;
    P.blkBitsA = cfg.blksw_bits_a; P.blkBitsB = cfg.blksw_bits_b;
    P.bitA = (unsigned)(1 - a / cfg.n_mdct_lines); P.bitB = (unsigned)(1 - b / cfg.n_mdct_lines);   // py2 int division
    const size_t wsBytes = pack_workspace_bytes(nChunks);
    MRC_HIP(h, h->packWs.reserve(wsBytes + (huff_table ? 0 : (size_t)nChunks * sizeof(int32_t))));
    int32_t* tableOut = huff_table ? huff_table : reinterpret_cast<int32_t*>(static_cast<char*>(h->packWs.p) + wsBytes);
    const int bound = (int)(mrc_pack_bound(&cfg, a, b, 1, joint) - 4);
    MRC_HIP(h, launch_pack(S, P, tables, n_blocks, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa,
                           mantissa_format, huff_table_in, tableOut, bits_saved, out, (long long)out_cap,
                           reinterpret_cast<long long*>(block_offset), h->packWs.p, bound, all_bands_non_empty(*hs), st));
    h->packLastChunks = nChunks;
    h->packLastCap = out_cap;
    if (total_bytes) {                                 // the caller wants the size now: one synchronisation
        long long total = 0;
        int bad = 0;
        MRC_HIP(h, hipMemcpyAsync(&total, pack_total_bytes(h->packWs.p, nChunks), sizeof(total), hipMemcpyDeviceToHost, st));
        MRC_HIP(h, hipMemcpyAsync(&bad, pack_error_flag(h->packWs.p, nChunks), sizeof(bad), hipMemcpyDeviceToHost, st));
        MRC_HIP(h, hipStreamSynchronize(st));
        *total_bytes = total;
        if (bad & 1) return fail(h, MRC_ERR_INVALID, "mrc_dev_pack_blocks: huff_table_in holds an id that is neither 0..3 nor 15");
        if (bad & 2)
            return fail(h, MRC_ERR_INVALID, "mrc_dev_pack_blocks: a chunk is larger than mrc_pack_bound allows (bit_alloc beyond "
                                            "16 bits?) and was not written");
        if (total > out_cap || (bad & 4)) return fail(h, MRC_ERR_NOMEM, "mrc_dev_pack_blocks: out_cap too small (see total_bytes)");
    }
    return MRC_OK;
}

int mrc_dev_pack_status(mrc_handle* h, int64_t* total_bytes, void* stream) {
    if (!h) return MRC_ERR_INVALID;
    if (total_bytes) *total_bytes = 0;
    if (!h->packLastChunks || !h->packWs.p) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    long long total = 0;
    int bad = 0;
    MRC_HIP(h, hipMemcpyAsync(&total, pack_total_bytes(h->packWs.p, h->packLastChunks), sizeof(total), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipMemcpyAsync(&bad, pack_error_flag(h->packWs.p, h->packLastChunks), sizeof(bad), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipStreamSynchronize(st));
    if (total_bytes) *total_bytes = total;
    if (bad & 3) return fail(h, MRC_ERR_INVALID, "mrc_dev_pack_blocks: bad table id or a chunk beyond mrc_pack_bound");
    if (total > h->packLastCap || (bad & 4)) return fail(h, MRC_ERR_NOMEM, "mrc_dev_pack_blocks: out_cap too small");
    return MRC_OK;
}

}  // extern "C"

namespace mrc {

// Phase A of the per-block path -- everything that does not depend on the bit reservoir: windowed MDCT + overall scale
// -> [M/S switch] -> SMRs and per-band peaks.  The M/S decision needs only the L / R lines (codecThem.py:436); made
// BEFORE the SMRs, it tells smr_kernel which of the four signals' SMRs the encoder will use per band
// (ms_stereo.py:70-81) -- the others are not computed.  Event order on the stream when timing: 0 | mdct | 1 | ms_switch
// | 2 | smr | 3.
int encode_phase_a(mrc_handle* h, const DevShape& S, int64_t n, const void* chL, const void* chR, int fmt, int64_t stride,
                   const int64_t* offsets, double* lines, int32_t* oscale, int32_t* msSwitch, double* smr, double* peak,
                   hipStream_t st, bool timing) {
    const int joint = chR ? 1 : 0;
    if (timing) MRC_HIP(h, hipEventRecord(h->ev[0], st));
    MRC_HIP(h, launch_mdct(S, n, chL, chR, fmt, stride, offsets, true, lines, oscale, st));
    if (timing) MRC_HIP(h, hipEventRecord(h->ev[1], st));
    if (joint)
        MRC_HIP(h, launch_ms_switch(n, S.nBands, S.msLeaves, S.msInternal, S.msPlan, lines, lines + S.halfN,
                                    4 * (int64_t)S.halfN, S.halfN, msSwitch, st));
    if (timing) MRC_HIP(h, hipEventRecord(h->ev[2], st));
    MRC_HIP(h, launch_smr(S, n, chL, chR, fmt, stride, offsets, lines, oscale, smr, nullptr, peak,
                          (joint && !h->smrAllBands) ? msSwitch : nullptr, h->exactSpread, st,
                          h->sensOn ? h->sens.as<unsigned long long>() : nullptr));
    if (timing) MRC_HIP(h, hipEventRecord(h->ev[3], st));
    return MRC_OK;
}

}  // namespace mrc

namespace {

// event order on the stream: 0 | mdct | 1 | ms_switch | 2 | smr | 3 | bitalloc (few blocks: event preparation) | 4 | quantize
// (few blocks: the scan kernel) | 5; reported order (mrc_get_kernel_ms): mdct, smr, ms_switch, bitalloc, quantize
int collect_kernel_ms(mrc_handle* h) {
    static const int from[5] = {0, 2, 1, 3, 4}, to[5] = {1, 3, 2, 4, 5};
    for (int i = 0; i < 5; ++i) {
        float ms = 0.f;
        MRC_HIP(h, hipEventElapsedTime(&ms, h->ev[from[i]], h->ev[to[i]]));
        h->kernelMs[i] = ms;
    }
    h->stageMs[0] = h->kernelMs[0];
    h->stageMs[1] = h->kernelMs[1];
    h->stageMs[2] = h->kernelMs[2] + h->kernelMs[3] + h->kernelMs[4];
    return MRC_OK;
}

// The whole per-block path for n blocks of one shape, queued on `st`: phase A, then bit allocation -> scale factors +
// mantissas.  Inputs and outputs are device pointers; `ws` holds the intermediate results and must not be shared with a
// call running on another stream.
int encode_core(mrc_handle* h, const DevShape& S, int64_t n, const void* chL, const void* chR, int fmt, int64_t stride,
                const int64_t* offsets, const int32_t* resIn, int32_t* oscale, int32_t* msSwitch, int32_t* bitAlloc,
                int32_t* scaleFactor, void* mantissa, int mantFmt, int32_t* resOut, double* linesOut, Workspace& ws,
                hipStream_t st) {
    const int joint = chR ? 1 : 0;
    const int nsig = joint ? 4 : 1;
    double* lines = linesOut;
    if (!lines) {
        MRC_HIP(h, ws.lines.reserve((size_t)n * nsig * S.halfN * sizeof(double)));
        lines = ws.lines.as<double>();
    }
    MRC_HIP(h, ws.smr.reserve((size_t)n * nsig * S.nBands * sizeof(double)));
    MRC_HIP(h, ws.peak.reserve(alloc_workspace_bytes(S, n, joint)));
    double* smr = ws.smr.as<double>();
    const bool timing = h->timing;
    MRC_TRY(encode_phase_a(h, S, n, chL, chR, fmt, stride, offsets, lines, oscale, msSwitch, smr, ws.peak.as<double>(), st,
                           timing));
    MRC_HIP(h, launch_alloc_quant(S, n, joint, lines, oscale, smr, resIn, msSwitch, bitAlloc, scaleFactor, mantissa,
                                  mantFmt, resOut, ws.peak.as<double>(), true, true, timing ? &h->ev[3] : nullptr, st));
    if (h->sensOn)                                       // (after the timed kernels; outside their events)
        MRC_HIP(h, launch_sensitivity(S, n, joint, lines, oscale, smr, ws.peak.as<double>(), msSwitch, bitAlloc, scaleFactor,
                                      h->sens.as<unsigned long long>(), nullptr, st));
    if (timing) {
        MRC_HIP(h, hipEventRecord(h->ev[5], st));
        MRC_HIP(h, hipEventSynchronize(h->ev[5]));
        MRC_TRY(collect_kernel_ms(h));
    }
    return MRC_OK;
}

}  // namespace

extern "C" {

int mrc_dev_encode_ex(mrc_handle* h, int a, int b, int64_t n_frames, const void* ch_left, const void* ch_right,
                      int sample_format, int64_t frame_stride, const int64_t* offsets, const int32_t* reservoir_in,
                      int32_t* overall_scale, int32_t* ms_switch, int32_t* bit_alloc, int32_t* scale_factor,
                      void* mantissa, int mantissa_format, int32_t* reservoir_out, double* lines_out, void* stream) {
    if (!h || !ch_left || !overall_scale || !bit_alloc || !scale_factor || !mantissa || !reservoir_out ||
        (ch_right && !ms_switch) || n_frames < 0 ||
        (sample_format != MRC_SAMPLES_F64 && sample_format != MRC_SAMPLES_PCM16) ||
        (mantissa_format != MRC_MANTISSA_I32 && mantissa_format != MRC_MANTISSA_I16))
        return fail(h, MRC_ERR_INVALID, "mrc_dev_encode: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    return encode_core(h, hs->dev, n_frames, ch_left, ch_right, sample_format, frame_stride, offsets, reservoir_in,
                       overall_scale, ms_switch, bit_alloc, scale_factor, mantissa, mantissa_format, reservoir_out,
                       lines_out, h->ws, pick_stream(h, stream));
}

int mrc_dev_encode(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                   int64_t frame_stride, const int64_t* offsets, const int32_t* reservoir_in,
                   int32_t* overall_scale, int32_t* ms_switch, int32_t* bit_alloc, int32_t* scale_factor,
                   int32_t* mantissa, int32_t* reservoir_out, double* lines_out, void* stream) {
    return mrc_dev_encode_ex(h, a, b, n_frames, ch_left, ch_right, MRC_SAMPLES_F64, frame_stride, offsets, reservoir_in,
                             overall_scale, ms_switch, bit_alloc, scale_factor, mantissa, MRC_MANTISSA_I32,
                             reservoir_out, lines_out, stream);
}

// ---------------------------------------------------------------------------------------------- host API

namespace {

struct Staged {
    mrc_handle* h;
    hipStream_t st;
    // Every host entry point owns one Staged: whichever way the function is left (also through an error return in
    // the middle), the stream is drained before the caller's buffers -- and any host temporaries declared BEFORE the
    // Staged object -- go out of scope under a copy that is still in flight.
    ~Staged() { if (st) (void)hipStreamSynchronize(st); }
    int up(DevBuf& buf, const void* src, size_t bytes) {
        MRC_HIP(h, buf.reserve(bytes ? bytes : 1));
        if (bytes) MRC_HIP(h, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
        return MRC_OK;
    }
    int down(void* dst, DevBuf& buf, size_t bytes) {
        if (bytes && dst) MRC_HIP(h, hipMemcpyAsync(dst, buf.p, bytes, hipMemcpyDeviceToHost, st));
        return MRC_OK;
    }
};

constexpr int64_t kSmallBatch = 64;               // calls with at most this many blocks go through one page-locked buffer each way

int encode_host(mrc_handle* h, int64_t n, int a, int b, const double* left, const double* right,
                const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch, int32_t* scale_factor,
                int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out, double* mdct_out) {
    if (!h || !left || !overall_scale || !scale_factor || !bit_alloc || !mantissa || !reservoir_out || n < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_encode: null output or negative block count");
    if (n == 0) return MRC_OK;
    const HostShape* hs;
    MRC_TRY(get_shape(h, a, b, &hs));
    const DevShape& S = hs->dev;
    const int joint = right ? 1 : 0, nsig = joint ? 4 : 1, nstream = joint ? 2 : 1;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    const size_t inBytes = (size_t)n * S.N * sizeof(double);
    const size_t szScale = (size_t)n * nsig * sizeof(int32_t), szSw = (size_t)n * S.nBands * sizeof(int32_t);
    const size_t szBand = (size_t)n * nstream * S.nBands * sizeof(int32_t);
    const size_t szMant = (size_t)n * nstream * S.halfN * sizeof(int32_t), szRes = (size_t)n * sizeof(int32_t);
    const size_t szLines = (size_t)n * nsig * S.halfN * sizeof(double);
    if (n <= kSmallBatch) {
        // The per-block seam (pacfileThem.py:649,820 -> codecThem.py:205-278 hands over ONE block per call).  Two things made
        // such a call slow (0.45 ms per joint block in round 3): every array in its own pageable copy (ten staging round trips
        // of the runtime), and the batch path's bit allocation -- one LANE per frame walking the greedy loop, ~135 us alone
        // when there is one frame.  Few blocks therefore take
        //   * ONE page-locked buffer each way and ONE device buffer laid out
        //       [left][right][scan descriptor][items][item starts] [reservoir] [scale][switch][sf][ba][mantissa16][lines]
        //     -- the copy in covers everything up to and including the reservoirs, the copy out everything from them on;
        //   * phase A as always, then the CHAINED encode's back end: the allocation as a sorted event list
        //     (chain_prep_kernel) and the scan kernel with every block as a stream of its own and Huffman pricing off, which
        //     leaves exactly what JointEncodeChannels / EncodeSingleChannel return (codecThem.py:332,503: the allocation's
        //     remainder as the reservoir; the savings are the caller's, 224,274).
        auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const int nEv = (int)chain_events_per_block(S, joint), nTot = nstream * S.nBands;
        const size_t szDesc = sizeof(ChainGroupDev), szItems = (size_t)n * sizeof(int32_t), szStart = (size_t)(n + 1) * sizeof(long long);
        const size_t szMant16 = (size_t)n * nstream * S.halfN * sizeof(uint16_t);
        const size_t oR = al(inBytes), oDesc = oR + (joint ? al(inBytes) : 0), oItems = oDesc + al(szDesc),
                     oStart = oItems + al(szItems), oRes = oStart + al(szStart), oScale = oRes + al(szRes),
                     oSw = oScale + al(szScale), oSf = oSw + al(joint ? szSw : 0), oBa = oSf + al(szBand), oMant = oBa + al(szBand),
                     oLines = oMant + al(szMant16), total = oLines + al(szLines);
        MRC_HIP(h, h->pinIn.reserve(oScale));
        MRC_HIP(h, h->pinOut.reserve(total - oRes));
        MRC_HIP(h, h->inL.reserve(total));
        MRC_HIP(h, h->inAux.reserve((size_t)n * nEv * sizeof(unsigned)));
        MRC_HIP(h, h->inAux2.reserve((size_t)n * (nEv + 1) * sizeof(unsigned)));
        MRC_HIP(h, h->outB.reserve((size_t)n * nstream * sizeof(int32_t)));          // table ids (all 15: no pricing)
        Workspace& ws = h->ws;
        MRC_HIP(h, ws.smr.reserve((size_t)n * nsig * S.nBands * sizeof(double)));
        MRC_HIP(h, ws.peak.reserve(alloc_workspace_bytes(S, n, joint)));
        char* pin = (char*)h->pinIn.p;
        char* dev = (char*)h->inL.p;
        std::memcpy(pin, left, inBytes);
        if (joint) std::memcpy(pin + oR, right, inBytes);
        ChainGroupDev D{};
        D.joint = joint; D.nb = S.nBands; D.nTot = nTot; D.M = S.halfN; D.K = S.maxMantBits - 1; D.nEv = nEv;
        D.nScaleBits = S.nScaleBits; D.nstream = nstream;
        D.maxN = 0;
        for (int v : hs->bandN) if (v > D.maxN) D.maxN = v;
        D.budgetMono = S.budgetMono; D.budgetJointPre = S.budgetJointPre; D.blkswA = S.blkswA; D.blkswB = S.blkswB;
        D.bandOfLine = S.bandOfLine; D.bandN = S.bandN;
        D.lines = (const double*)(dev + oLines); D.peak = ws.peak.as<double>(); D.oscale = (const int*)(dev + oScale);
        D.ms = (const int*)(dev + oSw); D.ev = h->inAux.as<unsigned>(); D.pre = h->inAux2.as<unsigned>();
        D.bitAlloc = (int*)(dev + oBa); D.scaleFactor = (int*)(dev + oSf); D.mant = (unsigned short*)(dev + oMant);
        D.table = h->outB.as<int32_t>();
        std::memcpy(pin + oDesc, &D, sizeof D);
        for (int64_t i = 0; i < n; ++i) {
            reinterpret_cast<int32_t*>(pin + oItems)[i] = (int32_t)i;                  // group 0, block i
            reinterpret_cast<long long*>(pin + oStart)[i] = i;
            reinterpret_cast<int32_t*>(pin + oRes)[i] = reservoir_in ? reservoir_in[i] : 0;
        }
        reinterpret_cast<long long*>(pin + oStart)[n] = n;
        hipStream_t st = h->stream;
        MRC_HIP(h, hipMemcpyAsync(dev, pin, oScale, hipMemcpyHostToDevice, st));
        const bool timing = h->timing;
        MRC_TRY(encode_phase_a(h, S, n, dev, joint ? dev + oR : nullptr, kSampleF64, S.N, nullptr, (double*)(dev + oLines),
                               (int32_t*)(dev + oScale), joint ? (int32_t*)(dev + oSw) : nullptr, ws.smr.as<double>(),
                               ws.peak.as<double>(), st, timing));
        MRC_HIP(h, launch_chain_prep(S, joint, n, ws.smr.as<double>(), joint ? (const int*)(dev + oSw) : nullptr,
                                     h->inAux.as<unsigned>(), h->inAux2.as<unsigned>(), 0, st));
        if (timing) MRC_HIP(h, hipEventRecord(h->ev[4], st));
        MRC_HIP(h, launch_chain_phase_b(n, (const ChainGroupDev*)(dev + oDesc), (const int*)(dev + oItems),
                                        (const long long*)(dev + oStart), (int*)(dev + oRes), nullptr, 0, h->chainThreads, st));
        if (timing) MRC_HIP(h, hipEventRecord(h->ev[5], st));
        if (h->sensOn)
            MRC_HIP(h, launch_sensitivity(S, n, joint, (const double*)(dev + oLines), (const int*)(dev + oScale),
                                          ws.smr.as<double>(), ws.peak.as<double>(), joint ? (const int*)(dev + oSw) : nullptr,
                                          (const int*)(dev + oBa), (const int*)(dev + oSf), h->sens.as<unsigned long long>(),
                                          nullptr, st));
        char* pout = (char*)h->pinOut.p;
        const size_t outBytes = (mdct_out ? total : oLines) - oRes;                     // (the lines only travel when asked for)
        MRC_HIP(h, hipMemcpyAsync(pout, dev + oRes, outBytes, hipMemcpyDeviceToHost, st));
        MRC_HIP(h, hipStreamSynchronize(st));
        if (timing) MRC_TRY(collect_kernel_ms(h));
        const char* o = pout - oRes;                                                    // (same offsets as on the device)
        std::memcpy(reservoir_out, o + oRes, szRes);
        std::memcpy(overall_scale, o + oScale, szScale);
        if (joint) std::memcpy(ms_switch, o + oSw, szSw);
        std::memcpy(scale_factor, o + oSf, szBand);
        std::memcpy(bit_alloc, o + oBa, szBand);
        const uint16_t* m16 = reinterpret_cast<const uint16_t*>(o + oMant);
        for (size_t i = 0; i < (size_t)n * nstream * S.halfN; ++i) mantissa[i] = (int32_t)m16[i];
        if (mdct_out) std::memcpy(mdct_out, o + oLines, szLines);
        return MRC_OK;
    }
    MRC_TRY(s.up(h->inL, left, inBytes));
    if (joint) MRC_TRY(s.up(h->inR, right, inBytes));
    if (reservoir_in) MRC_TRY(s.up(h->inAux, reservoir_in, (size_t)n * sizeof(int32_t)));
    MRC_HIP(h, h->outA.reserve(szScale)); MRC_HIP(h, h->outB.reserve(szSw)); MRC_HIP(h, h->outC.reserve(szBand));
    MRC_HIP(h, h->outD.reserve(szBand));  MRC_HIP(h, h->outE.reserve(szMant)); MRC_HIP(h, h->outF.reserve(szRes));
    MRC_HIP(h, h->outG.reserve(szLines));
    MRC_TRY(mrc_dev_encode(h, a, b, n, h->inL.as<double>(), joint ? h->inR.as<double>() : nullptr, S.N, nullptr,
                           reservoir_in ? h->inAux.as<int32_t>() : nullptr, h->outA.as<int32_t>(),
                           h->outB.as<int32_t>(), h->outD.as<int32_t>(), h->outC.as<int32_t>(), h->outE.as<int32_t>(),
                           h->outF.as<int32_t>(), h->outG.as<double>(), h->stream));
    MRC_TRY(s.down(overall_scale, h->outA, szScale));
    if (joint) MRC_TRY(s.down(ms_switch, h->outB, szSw));
    MRC_TRY(s.down(scale_factor, h->outC, szBand));
    MRC_TRY(s.down(bit_alloc, h->outD, szBand));
    MRC_TRY(s.down(mantissa, h->outE, szMant));
    MRC_TRY(s.down(reservoir_out, h->outF, szRes));
    MRC_TRY(s.down(mdct_out, h->outG, szLines));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

}  // namespace

int mrc_encode_mono(mrc_handle* h, int64_t n_blocks, int a, int b, const double* blocks, const int32_t* reservoir_in,
                    int32_t* overall_scale, int32_t* scale_factor, int32_t* bit_alloc, int32_t* mantissa,
                    int32_t* reservoir_out, double* mdct_out) {
    return encode_host(h, n_blocks, a, b, blocks, nullptr, reservoir_in, overall_scale, nullptr, scale_factor,
                       bit_alloc, mantissa, reservoir_out, mdct_out);
}

int mrc_encode_joint(mrc_handle* h, int64_t n_blocks, int a, int b, const double* left, const double* right,
                     const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch, int32_t* scale_factor,
                     int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out, double* mdct_out) {
    if (!right || !ms_switch) return fail(h, MRC_ERR_INVALID, "mrc_encode_joint: null right channel or ms_switch");
    return encode_host(h, n_blocks, a, b, left, right, reservoir_in, overall_scale, ms_switch, scale_factor, bit_alloc,
                       mantissa, reservoir_out, mdct_out);
}

// ---- pinned host memory: what makes the copies of mrc_encode_stream_pcm16 asynchronous --------------------------
int mrc_host_alloc(void** out, size_t bytes) {
    if (!out) return MRC_ERR_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? MRC_OK : MRC_ERR_NOMEM;
}
int mrc_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? MRC_OK : MRC_ERR_HIP; }
int mrc_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return MRC_ERR_INVALID;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? MRC_OK : MRC_ERR_HIP;
}
int mrc_host_unregister(void* p) { return (!p || hipHostUnregister(p) == hipSuccess) ? MRC_OK : MRC_ERR_HIP; }

// ---- 16-bit PCM in host memory -> codes in host memory, pipelined ---------------------------------------------
int mrc_encode_stream_pcm16(mrc_handle* h, int64_t n_frames, const int16_t* pcm_left, const int16_t* pcm_right,
                            const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch,
                            int32_t* scale_factor, int32_t* bit_alloc, uint16_t* mantissa16, int32_t* reservoir_out,
                            int64_t chunk_frames) {
    if (!h || !pcm_left || !overall_scale || !scale_factor || !bit_alloc || !mantissa16 || !reservoir_out ||
        (pcm_right && !ms_switch) || n_frames < 0 || chunk_frames < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_encode_stream_pcm16: bad argument");
    if (n_frames == 0) return MRC_OK;
    const int L = h->cfg.n_mdct_lines;
    const HostShape* hs;
    MRC_TRY(get_shape(h, L, L, &hs));
    const DevShape& S = hs->dev;
    const int joint = pcm_right ? 1 : 0, nsig = joint ? 4 : 1, nstream = joint ? 2 : 1;
    // default chunk: 32 768 frames (64 MiB each way: below that the fixed cost of a copy shows -- 9 000 Msamples/s at
    // 8 192 frames against 19 000 at 32 768), halved while the stream has fewer than six chunks to pipeline
    int64_t chunk = chunk_frames;
    if (!chunk)
        for (chunk = 32768; chunk > 8192 && n_frames < 6 * chunk;) chunk /= 2;
    if (chunk > n_frames) chunk = n_frames;
    // streams, events and chunk buffers: created on first use, buffers sized for one chunk (+ the one-hop halo in front)
    for (hipStream_t* st : {&h->stIn, &h->stOut})
        if (!*st) MRC_HIP(h, hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    const hipStream_t stK = h->stream;
    const size_t szPcm = (size_t)(chunk + 1) * L * sizeof(int16_t);
    for (auto& lane : h->lanes) {
        for (hipEvent_t* e : {&lane.evIn, &lane.evK, &lane.evOut})
            if (!*e) MRC_HIP(h, hipEventCreateWithFlags(e, hipEventDisableTiming));
        MRC_HIP(h, lane.pcmL.reserve(szPcm));
        if (joint) MRC_HIP(h, lane.pcmR.reserve(szPcm));
        MRC_HIP(h, lane.resIn.reserve((size_t)chunk * sizeof(int32_t)));
        MRC_HIP(h, lane.oScale.reserve((size_t)chunk * nsig * sizeof(int32_t)));
        MRC_HIP(h, lane.ms.reserve((size_t)chunk * S.nBands * sizeof(int32_t)));
        MRC_HIP(h, lane.ba.reserve((size_t)chunk * nstream * S.nBands * sizeof(int32_t)));
        MRC_HIP(h, lane.sf.reserve((size_t)chunk * nstream * S.nBands * sizeof(int32_t)));
        MRC_HIP(h, lane.mant.reserve((size_t)chunk * nstream * S.halfN * sizeof(uint16_t)));
        MRC_HIP(h, lane.resOut.reserve((size_t)chunk * sizeof(int32_t)));
    }
    MRC_HIP(h, h->wsPipe.lines.reserve((size_t)chunk * nsig * S.halfN * sizeof(double)));
    MRC_HIP(h, h->wsPipe.smr.reserve((size_t)chunk * nsig * S.nBands * sizeof(double)));
    MRC_HIP(h, h->wsPipe.peak.reserve(alloc_workspace_bytes(S, chunk, joint)));
    struct DrainAll {                       // whichever way we leave: nothing of ours is still using the caller's memory
        mrc_handle* h;
        ~DrainAll() { for (hipStream_t st : {h->stIn, h->stream, h->stOut}) if (st) (void)hipStreamSynchronize(st); }
    } drain{h};
    const bool wasTiming = h->timing;
    h->timing = false;                      // the per-kernel events of encode_core belong to ONE call at a time
    int rc = MRC_OK;
    int64_t c = 0;
    for (int64_t f0 = 0; f0 < n_frames && rc == MRC_OK; f0 += chunk, ++c) {
        Lane& lane = h->lanes[c % kLanes];
        const bool reused = c >= kLanes;    // the lane's events still stand for chunk c - kLanes when they are waited on here
        const int64_t n = (n_frames - f0 < chunk) ? n_frames - f0 : chunk;
        const size_t inBytes = (size_t)(n + 1) * L * sizeof(int16_t);
#define MRC_Q(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = hip_fail(h, e_, #call); break; } } while (0)
        do {
            // copy in (the lane's input buffers are free once the kernels of its previous chunk have run)
            if (reused) MRC_Q(hipStreamWaitEvent(h->stIn, lane.evK, 0));
            MRC_Q(hipMemcpyAsync(lane.pcmL.p, pcm_left + f0 * L, inBytes, hipMemcpyHostToDevice, h->stIn));
            if (joint) MRC_Q(hipMemcpyAsync(lane.pcmR.p, pcm_right + f0 * L, inBytes, hipMemcpyHostToDevice, h->stIn));
            if (reservoir_in)
                MRC_Q(hipMemcpyAsync(lane.resIn.p, reservoir_in + f0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stIn));
            MRC_Q(hipEventRecord(lane.evIn, h->stIn));
            // kernels (the lane's output buffers are free once its previous chunk has been copied out)
            MRC_Q(hipStreamWaitEvent(stK, lane.evIn, 0));
            if (reused) MRC_Q(hipStreamWaitEvent(stK, lane.evOut, 0));
            rc = encode_core(h, S, n, lane.pcmL.p, joint ? lane.pcmR.p : nullptr, kSampleI16, L, nullptr,
                             reservoir_in ? lane.resIn.as<int32_t>() : nullptr, lane.oScale.as<int32_t>(),
                             lane.ms.as<int32_t>(), lane.ba.as<int32_t>(), lane.sf.as<int32_t>(), lane.mant.p,
                             MRC_MANTISSA_I16, lane.resOut.as<int32_t>(), nullptr, h->wsPipe, stK);
            if (rc != MRC_OK) break;
            MRC_Q(hipEventRecord(lane.evK, stK));
            // copy out
            hipStream_t so = h->stOut;
            MRC_Q(hipStreamWaitEvent(so, lane.evK, 0));
            MRC_Q(hipMemcpyAsync(mantissa16 + f0 * nstream * S.halfN, lane.mant.p, (size_t)n * nstream * S.halfN * sizeof(uint16_t), hipMemcpyDeviceToHost, so));
            MRC_Q(hipMemcpyAsync(overall_scale + f0 * nsig, lane.oScale.p, (size_t)n * nsig * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            if (joint)
                MRC_Q(hipMemcpyAsync(ms_switch + f0 * S.nBands, lane.ms.p, (size_t)n * S.nBands * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            MRC_Q(hipMemcpyAsync(bit_alloc + f0 * nstream * S.nBands, lane.ba.p, (size_t)n * nstream * S.nBands * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            MRC_Q(hipMemcpyAsync(scale_factor + f0 * nstream * S.nBands, lane.sf.p, (size_t)n * nstream * S.nBands * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            MRC_Q(hipMemcpyAsync(reservoir_out + f0, lane.resOut.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            MRC_Q(hipEventRecord(lane.evOut, so));
        } while (0);
#undef MRC_Q
    }
    h->timing = wasTiming;
    if (rc != MRC_OK) return rc;
    for (hipStream_t st : {h->stIn, stK, h->stOut}) MRC_HIP(h, hipStreamSynchronize(st));
    return MRC_OK;
}

// ---- 16-bit PCM in host memory -> `.pac` chunk bytes in host memory, pipelined ---------------------------------
// mrc_encode_stream_pcm16 with the back end of mrc_dev_pack_blocks behind the kernels of every chunk: what comes back
// over PCIe is the packed chunks (a few hundred bytes per frame and channel) instead of the mantissa plane.  The size of
// a chunk's packed form is only known once its pack kernels have run: the host reads it (8 bytes, page-locked) one chunk
// BEHIND the chunk it is queueing, so the wait never leaves the device idle.
int mrc_encode_stream_pcm16_pac(mrc_handle* h, int64_t n_frames, const int16_t* pcm_left, const int16_t* pcm_right,
                                const int32_t* reservoir_in, int use_huffman, uint8_t* out, int64_t out_cap,
                                int64_t* block_offset, int32_t* huff_table, int32_t* bits_saved, int32_t* reservoir_out,
                                int64_t* total_bytes, int64_t chunk_frames) {
    if (!h || !pcm_left || !out || !block_offset || !total_bytes || n_frames < 0 || chunk_frames < 0 || out_cap < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_encode_stream_pcm16_pac: bad argument");
    *total_bytes = 0;
    block_offset[0] = 0;
    if (n_frames == 0) return MRC_OK;
    const int L = h->cfg.n_mdct_lines;
    const HostShape* hs;
    MRC_TRY(get_shape(h, L, L, &hs));
    const DevShape& S = hs->dev;
    const mrc_config& cfg = h->cfg;
    const int joint = pcm_right ? 1 : 0, nsig = joint ? 4 : 1, nch = joint ? 2 : 1;
    int64_t chunk = chunk_frames;
    if (!chunk)
        for (chunk = 32768; chunk > 8192 && n_frames < 6 * chunk;) chunk /= 2;
    if (chunk > n_frames) chunk = n_frames;
    const int64_t bound = mrc_pack_bound(&cfg, L, L, 1, joint);             // worst case per channel chunk, length field included
    if (bound < 0) return fail(h, MRC_ERR_INVALID, "mrc_encode_stream_pcm16_pac: field widths out of range");
    for (hipStream_t* st : {&h->stIn, &h->stOut})
        if (!*st) MRC_HIP(h, hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    const hipStream_t stK = h->stream;
    const size_t szPcm = (size_t)(chunk + 1) * L * sizeof(int16_t);
    const size_t pacCap = (size_t)chunk * nch * (size_t)bound;
    for (auto& lane : h->lanes) {
        for (hipEvent_t* e : {&lane.evIn, &lane.evK, &lane.evOut})
            if (!*e) MRC_HIP(h, hipEventCreateWithFlags(e, hipEventDisableTiming));
        if (!lane.pacTotal) {
            MRC_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&lane.pacTotal), 2 * sizeof(long long), hipHostMallocDefault));
            lane.pacTotal[0] = lane.pacTotal[1] = 0;
        }
        MRC_HIP(h, lane.pcmL.reserve(szPcm));
        if (joint) MRC_HIP(h, lane.pcmR.reserve(szPcm));
        MRC_HIP(h, lane.resIn.reserve((size_t)chunk * sizeof(int32_t)));
        MRC_HIP(h, lane.oScale.reserve((size_t)chunk * nsig * sizeof(int32_t)));
        MRC_HIP(h, lane.ms.reserve((size_t)chunk * S.nBands * sizeof(int32_t)));
        MRC_HIP(h, lane.ba.reserve((size_t)chunk * nch * S.nBands * sizeof(int32_t)));
        MRC_HIP(h, lane.sf.reserve((size_t)chunk * nch * S.nBands * sizeof(int32_t)));
        MRC_HIP(h, lane.mant.reserve((size_t)chunk * nch * S.halfN * sizeof(uint16_t)));
        MRC_HIP(h, lane.resOut.reserve((size_t)chunk * sizeof(int32_t)));
        MRC_HIP(h, lane.pacBytes.reserve(pacCap));
        MRC_HIP(h, lane.pacOffs.reserve((size_t)(chunk + 1) * sizeof(int64_t)));
        MRC_HIP(h, lane.pacTable.reserve((size_t)chunk * nch * sizeof(int32_t)));
        MRC_HIP(h, lane.pacSaved.reserve((size_t)chunk * nch * sizeof(int32_t)));
    }
    MRC_HIP(h, h->wsPipe.lines.reserve((size_t)chunk * nsig * S.halfN * sizeof(double)));
    MRC_HIP(h, h->wsPipe.smr.reserve((size_t)chunk * nsig * S.nBands * sizeof(double)));
    MRC_HIP(h, h->wsPipe.peak.reserve(alloc_workspace_bytes(S, chunk, joint)));
    struct DrainAll {
        mrc_handle* h;
        ~DrainAll() { for (hipStream_t st : {h->stIn, h->stream, h->stOut}) if (st) (void)hipStreamSynchronize(st); }
    } drain{h};
    const bool wasTiming = h->timing;
    h->timing = false;
    int rc = MRC_OK;
    const int64_t nChunks = (n_frames + chunk - 1) / chunk;
    std::vector<int64_t> base((size_t)nChunks + 1, 0);                      // where every chunk's bytes start in `out`
#define MRC_Q(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = hip_fail(h, e_, #call); break; } } while (0)
    // copies chunk c's packed form out, once its size is known (called one chunk behind the queueing loop)
    auto collect = [&](int64_t c) {
        Lane& lane = h->lanes[c % kLanes];
        const int64_t f0 = c * chunk;
        const int64_t n = (n_frames - f0 < chunk) ? n_frames - f0 : chunk;
        do {
            MRC_Q(hipEventSynchronize(lane.evK));
            const long long total = lane.pacTotal[0];
            if (lane.pacTotal[1] & 3) { rc = fail(h, MRC_ERR_INVALID, "mrc_encode_stream_pcm16_pac: internal error (table id / chunk size out of range)"); break; }
            base[(size_t)c + 1] = base[(size_t)c] + total;
            if (base[(size_t)c + 1] > out_cap) { rc = fail(h, MRC_ERR_NOMEM, "mrc_encode_stream_pcm16_pac: out_cap too small"); break; }
            hipStream_t so = h->stOut;
            MRC_Q(hipMemcpyAsync(out + base[(size_t)c], lane.pacBytes.p, (size_t)total, hipMemcpyDeviceToHost, so));
            MRC_Q(hipMemcpyAsync(block_offset + f0, lane.pacOffs.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, so));
            if (huff_table)
                MRC_Q(hipMemcpyAsync(huff_table + f0 * nch, lane.pacTable.p, (size_t)n * nch * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            if (bits_saved)
                MRC_Q(hipMemcpyAsync(bits_saved + f0 * nch, lane.pacSaved.p, (size_t)n * nch * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            if (reservoir_out)
                MRC_Q(hipMemcpyAsync(reservoir_out + f0, lane.resOut.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, so));
            MRC_Q(hipEventRecord(lane.evOut, so));
        } while (0);
    };
    constexpr int64_t kCollectLag = 2;      // < kLanes - 1: chunk c's buffers are reused by chunk c + kLanes
    static_assert(kCollectLag < kLanes - 1, "a lane must be collected before it is queued again");
    static const PackTables tables = [] { PackTables t; pack_tables(&t); return t; }();
    PackParams P;
    P.nch = nch; P.joint = joint; P.useHuffman = use_huffman ? 1 : 0;
    P.nScaleBits = cfg.n_scale_bits; P.nMantSizeBits = cfg.n_mant_size_bits;
    P.blkBitsA = cfg.blksw_bits_a; P.blkBitsB = cfg.blksw_bits_b;
    P.bitA = 0; P.bitB = 0;                                                 // long blocks: 1 - a / nMDCTLines = 0
    MRC_HIP(h, h->packWs.reserve(pack_workspace_bytes(chunk * nch)));
    for (int64_t c = 0; c < nChunks && rc == MRC_OK; ++c) {
        Lane& lane = h->lanes[c % kLanes];
        const bool reused = c >= kLanes;
        const int64_t f0 = c * chunk;
        const int64_t n = (n_frames - f0 < chunk) ? n_frames - f0 : chunk;
        const size_t inBytes = (size_t)(n + 1) * L * sizeof(int16_t);
        do {
            if (reused) MRC_Q(hipStreamWaitEvent(h->stIn, lane.evK, 0));
            MRC_Q(hipMemcpyAsync(lane.pcmL.p, pcm_left + f0 * L, inBytes, hipMemcpyHostToDevice, h->stIn));
            if (joint) MRC_Q(hipMemcpyAsync(lane.pcmR.p, pcm_right + f0 * L, inBytes, hipMemcpyHostToDevice, h->stIn));
            if (reservoir_in)
                MRC_Q(hipMemcpyAsync(lane.resIn.p, reservoir_in + f0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stIn));
            MRC_Q(hipEventRecord(lane.evIn, h->stIn));
            MRC_Q(hipStreamWaitEvent(stK, lane.evIn, 0));
            if (reused) MRC_Q(hipStreamWaitEvent(stK, lane.evOut, 0));      // (chunk c - kLanes was collected below, kLanes - kCollectLag turns ago)
            rc = encode_core(h, S, n, lane.pcmL.p, joint ? lane.pcmR.p : nullptr, kSampleI16, L, nullptr,
                             reservoir_in ? lane.resIn.as<int32_t>() : nullptr, lane.oScale.as<int32_t>(),
                             lane.ms.as<int32_t>(), lane.ba.as<int32_t>(), lane.sf.as<int32_t>(), lane.mant.p,
                             MRC_MANTISSA_I16, lane.resOut.as<int32_t>(), nullptr, h->wsPipe, stK);
            if (rc != MRC_OK) break;
            MRC_Q(launch_pack(S, P, tables, n, lane.oScale.as<int32_t>(), lane.ms.as<int32_t>(), lane.sf.as<int32_t>(),
                              lane.ba.as<int32_t>(), lane.mant.p, MRC_MANTISSA_I16, nullptr, lane.pacTable.as<int32_t>(),
                              lane.pacSaved.as<int32_t>(), lane.pacBytes.as<unsigned char>(), (long long)pacCap,
                              lane.pacOffs.as<long long>(), h->packWs.p, (int)(bound - 4), all_bands_non_empty(*hs), stK));
            MRC_Q(launch_pack_export(h->packWs.p, n * nch, lane.pacTotal, stK));   // 16 bytes, written by a kernel: no copy command
            MRC_Q(hipEventRecord(lane.evK, stK));
        } while (0);
        if (rc == MRC_OK && c >= kCollectLag) collect(c - kCollectLag);
    }
    for (int64_t c = nChunks > kCollectLag ? nChunks - kCollectLag : 0; c < nChunks && rc == MRC_OK; ++c) collect(c);
#undef MRC_Q
    h->timing = wasTiming;
    if (rc != MRC_OK) return rc;
    for (hipStream_t st : {h->stIn, stK, h->stOut}) MRC_HIP(h, hipStreamSynchronize(st));
    // the chunks' block offsets count from their own first byte
    for (int64_t c = 0; c < nChunks; ++c) {
        const int64_t f0 = c * chunk, n = (n_frames - f0 < chunk) ? n_frames - f0 : chunk;
        if (base[(size_t)c])
            for (int64_t i = 0; i < n; ++i) block_offset[f0 + i] += base[(size_t)c];
    }
    block_offset[n_frames] = base[(size_t)nChunks];
    *total_bytes = base[(size_t)nChunks];
    return MRC_OK;
}

// ---- blocks of MIXED shapes in one call (a block-switched stream): grouped by shape, one launch set per shape ------
namespace {

int encode_blocks_host(mrc_handle* h, int64_t n, const double* left, const double* right, const int32_t* a,
                       const int32_t* b, const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch,
                       int32_t* scale_factor, int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out) {
    if (!h || !left || !a || !b || !overall_scale || !scale_factor || !bit_alloc || !mantissa || !reservoir_out ||
        (right && !ms_switch) || n < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_encode_blocks: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    const int joint = right ? 1 : 0, nsig = joint ? 4 : 1, nstream = joint ? 2 : 1;
    const int L = h->cfg.n_mdct_lines;
    // where each block starts in the packed input, and which blocks share a shape
    std::vector<int64_t> start((size_t)n + 1);
    std::map<std::pair<int, int>, std::vector<int64_t>> groups;
    start[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (a[i] <= 0 || b[i] <= 0 || (a[i] + b[i]) / 2 > L)
            return fail(h, MRC_ERR_INVALID, "mrc_encode_blocks: block shape out of range (a, b > 0, (a+b)/2 <= n_mdct_lines)");
        start[(size_t)i + 1] = start[(size_t)i] + a[i] + b[i];
        groups[std::make_pair((int)a[i], (int)b[i])].push_back(i);
    }
    std::vector<int64_t> offs;
    std::vector<int32_t> resG, tScale, tSw, tSf, tBa, tMant, tRes;
    Staged s{h, h->stream};                                 // (the vectors above outlive its drain)
    const size_t inBytes = (size_t)start[(size_t)n] * sizeof(double);
    MRC_TRY(s.up(h->inL, left, inBytes));
    if (joint) MRC_TRY(s.up(h->inR, right, inBytes));
    for (auto& kv : groups) {
        const int ga = kv.first.first, gb = kv.first.second;
        const std::vector<int64_t>& idx = kv.second;
        const int64_t m = (int64_t)idx.size();
        const HostShape* hs;
        MRC_TRY(get_shape(h, ga, gb, &hs));
        const DevShape& S = hs->dev;
        offs.resize((size_t)m);
        resG.assign((size_t)m, 0);
        for (int64_t j = 0; j < m; ++j) {
            offs[(size_t)j] = start[(size_t)idx[(size_t)j]];
            if (reservoir_in) resG[(size_t)j] = reservoir_in[idx[(size_t)j]];
        }
        const size_t szScale = (size_t)m * nsig * sizeof(int32_t), szSw = (size_t)m * S.nBands * sizeof(int32_t);
        const size_t szBand = (size_t)m * nstream * S.nBands * sizeof(int32_t);
        const size_t szMant = (size_t)m * nstream * S.halfN * sizeof(int32_t), szRes = (size_t)m * sizeof(int32_t);
        MRC_TRY(s.up(h->inAux, offs.data(), (size_t)m * sizeof(int64_t)));
        MRC_TRY(s.up(h->inAux2, resG.data(), szRes));
        MRC_HIP(h, h->outA.reserve(szScale)); MRC_HIP(h, h->outB.reserve(szSw)); MRC_HIP(h, h->outC.reserve(szBand));
        MRC_HIP(h, h->outD.reserve(szBand));  MRC_HIP(h, h->outE.reserve(szMant)); MRC_HIP(h, h->outF.reserve(szRes));
        MRC_TRY(encode_core(h, S, m, h->inL.p, joint ? h->inR.p : nullptr, kSampleF64, 0, h->inAux.as<int64_t>(),
                            h->inAux2.as<int32_t>(), h->outA.as<int32_t>(), h->outB.as<int32_t>(), h->outD.as<int32_t>(),
                            h->outC.as<int32_t>(), h->outE.p, MRC_MANTISSA_I32, h->outF.as<int32_t>(), nullptr, h->ws,
                            h->stream));
        tScale.resize((size_t)m * nsig); tSw.resize((size_t)m * S.nBands); tSf.resize((size_t)m * nstream * S.nBands);
        tBa.resize((size_t)m * nstream * S.nBands); tMant.resize((size_t)m * nstream * S.halfN); tRes.resize((size_t)m);
        MRC_TRY(s.down(tScale.data(), h->outA, szScale));
        if (joint) MRC_TRY(s.down(tSw.data(), h->outB, szSw));
        MRC_TRY(s.down(tSf.data(), h->outC, szBand));
        MRC_TRY(s.down(tBa.data(), h->outD, szBand));
        MRC_TRY(s.down(tMant.data(), h->outE, szMant));
        MRC_TRY(s.down(tRes.data(), h->outF, szRes));
        MRC_HIP(h, hipStreamSynchronize(h->stream));
        // scatter into the caller's fixed-stride arrays (rows beyond the shape's band / line count are zeroed)
        for (int64_t j = 0; j < m; ++j) {
            const int64_t i = idx[(size_t)j];
            for (int g = 0; g < nsig; ++g) overall_scale[i * nsig + g] = tScale[(size_t)(j * nsig + g)];
            reservoir_out[i] = tRes[(size_t)j];
            if (joint)
                for (int k = 0; k < MRC_MAX_BANDS; ++k)
                    ms_switch[i * MRC_MAX_BANDS + k] = k < S.nBands ? tSw[(size_t)(j * S.nBands + k)] : 0;
            for (int c = 0; c < nstream; ++c) {
                int32_t* sfRow = scale_factor + (i * nstream + c) * MRC_MAX_BANDS;
                int32_t* baRow = bit_alloc + (i * nstream + c) * MRC_MAX_BANDS;
                int32_t* mRow = mantissa + (i * nstream + c) * (int64_t)L;
                for (int k = 0; k < MRC_MAX_BANDS; ++k) {
                    sfRow[k] = k < S.nBands ? tSf[(size_t)((j * nstream + c) * S.nBands + k)] : 0;
                    baRow[k] = k < S.nBands ? tBa[(size_t)((j * nstream + c) * S.nBands + k)] : 0;
                }
                std::memcpy(mRow, &tMant[(size_t)((j * nstream + c) * S.halfN)], (size_t)S.halfN * sizeof(int32_t));
                std::memset(mRow + S.halfN, 0, (size_t)(L - S.halfN) * sizeof(int32_t));
            }
        }
    }
    return MRC_OK;
}

}  // namespace

int mrc_encode_mono_blocks(mrc_handle* h, int64_t n_blocks, const double* blocks, const int32_t* a, const int32_t* b,
                           const int32_t* reservoir_in, int32_t* overall_scale, int32_t* scale_factor,
                           int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out) {
    return encode_blocks_host(h, n_blocks, blocks, nullptr, a, b, reservoir_in, overall_scale, nullptr, scale_factor,
                              bit_alloc, mantissa, reservoir_out);
}

int mrc_encode_joint_blocks(mrc_handle* h, int64_t n_blocks, const double* left, const double* right, const int32_t* a,
                            const int32_t* b, const int32_t* reservoir_in, int32_t* overall_scale, int32_t* ms_switch,
                            int32_t* scale_factor, int32_t* bit_alloc, int32_t* mantissa, int32_t* reservoir_out) {
    if (!right || !ms_switch) return fail(h, MRC_ERR_INVALID, "mrc_encode_joint_blocks: null right channel or ms_switch");
    return encode_blocks_host(h, n_blocks, left, right, a, b, reservoir_in, overall_scale, ms_switch, scale_factor,
                              bit_alloc, mantissa, reservoir_out);
}

int mrc_quantize_uniform(mrc_handle* h, int64_t n, int n_bits, const double* x, int64_t* code) {
    if (!h || n < 0 || !x || !code || n_bits < 1 || n_bits > 62)
        return fail(h, MRC_ERR_INVALID, "mrc_quantize_uniform: bad argument (1 <= n_bits <= 62)");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, x, (size_t)n * sizeof(double)));
    MRC_HIP(h, h->outG.reserve((size_t)n * sizeof(int64_t)));
    MRC_HIP(h, launch_quantize_uniform(n, n_bits, h->inL.as<double>(), h->outG.as<long long>(), h->stream));
    MRC_TRY(s.down(code, h->outG, (size_t)n * sizeof(int64_t)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_bark(mrc_handle* h, int64_t n, const double* f, double* z) {
    if (!h || n < 0 || !f || !z) return fail(h, MRC_ERR_INVALID, "mrc_bark: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, f, (size_t)n * sizeof(double)));
    MRC_HIP(h, h->outG.reserve((size_t)n * sizeof(double)));
    MRC_HIP(h, launch_bark(n, h->inL.as<double>(), h->outG.as<double>(), h->stream));
    MRC_TRY(s.down(z, h->outG, (size_t)n * sizeof(double)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_pcm_to_float(mrc_handle* h, int64_t n, const int16_t* pcm, double* out) {
    if (!h || n < 0 || !pcm || !out) return fail(h, MRC_ERR_INVALID, "mrc_pcm_to_float: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, pcm, (size_t)n * sizeof(int16_t)));
    MRC_HIP(h, h->outG.reserve((size_t)n * sizeof(double)));
    MRC_HIP(h, launch_pcm_to_float(n, h->inL.as<short>(), h->outG.as<double>(), h->stream));
    MRC_TRY(s.down(out, h->outG, (size_t)n * sizeof(double)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_window(mrc_handle* h, int64_t n, int a, int b, const double* blocks, double* out) {
    if (!h || !blocks || !out || n < 0) return fail(h, MRC_ERR_INVALID, "mrc_window: bad argument");
    if (n == 0) return MRC_OK;
    const HostShape* hs;
    MRC_TRY(get_shape(h, a, b, &hs));
    const DevShape& S = hs->dev;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    const size_t bytes = (size_t)n * S.N * sizeof(double);
    MRC_TRY(s.up(h->inL, blocks, bytes));
    MRC_HIP(h, h->outG.reserve(bytes));
    MRC_HIP(h, launch_window(S, n, h->inL.as<double>(), h->outG.as<double>(), h->stream));
    MRC_TRY(s.down(out, h->outG, bytes));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_mdct(mrc_handle* h, int64_t n, int a, int b, const double* blocks, int apply_window, double* lines,
             int32_t* overall_scale) {
    if (!h || !blocks || !lines || !overall_scale || n < 0) return fail(h, MRC_ERR_INVALID, "mrc_mdct: bad argument");
    if (n == 0) return MRC_OK;
    const HostShape* hs;
    MRC_TRY(get_shape(h, a, b, &hs));
    const DevShape& S = hs->dev;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, blocks, (size_t)n * S.N * sizeof(double)));
    const size_t szLines = (size_t)n * S.halfN * sizeof(double), szScale = (size_t)n * sizeof(int32_t);
    MRC_HIP(h, h->outG.reserve(szLines)); MRC_HIP(h, h->outA.reserve(szScale));
    MRC_HIP(h, launch_mdct(S, n, h->inL.as<double>(), nullptr, kSampleF64, S.N, nullptr, apply_window != 0, h->outG.as<double>(),
                           h->outA.as<int>(), h->stream));
    MRC_TRY(s.down(lines, h->outG, szLines));
    MRC_TRY(s.down(overall_scale, h->outA, szScale));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_smr(mrc_handle* h, int64_t n, int a, int b, const double* blocks, const double* scaled_lines,
            const int32_t* overall_scale, double* smr, double* thresh) {
    if (!h || !blocks || !smr || n < 0 || ((scaled_lines == nullptr) != (overall_scale == nullptr)))
        return fail(h, MRC_ERR_INVALID, "mrc_smr: bad argument (scaled_lines and overall_scale go together)");
    if (n == 0) return MRC_OK;
    const HostShape* hs;
    MRC_TRY(get_shape(h, a, b, &hs));
    const DevShape& S = hs->dev;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, blocks, (size_t)n * S.N * sizeof(double)));
    const size_t szLines = (size_t)n * S.halfN * sizeof(double), szScale = (size_t)n * sizeof(int32_t);
    const size_t szSmr = (size_t)n * S.nBands * sizeof(double);
    MRC_HIP(h, h->outG.reserve(szLines)); MRC_HIP(h, h->outA.reserve(szScale));
    MRC_HIP(h, h->outC.reserve(szSmr)); MRC_HIP(h, h->outE.reserve(szLines));
    if (scaled_lines) {
        MRC_TRY(s.up(h->inR, scaled_lines, szLines));
        MRC_TRY(s.up(h->outA, overall_scale, szScale));
        MRC_HIP(h, launch_unscale(n, S.halfN, h->inR.as<double>(), h->outA.as<int>(), h->outG.as<double>(), h->stream));
    } else {
        MRC_HIP(h, launch_mdct(S, n, h->inL.as<double>(), nullptr, kSampleF64, S.N, nullptr, true, h->outG.as<double>(),
                               h->outA.as<int>(), h->stream));
    }
    MRC_HIP(h, launch_smr(S, n, h->inL.as<double>(), nullptr, kSampleF64, S.N, nullptr, h->outG.as<double>(), h->outA.as<int>(),
                          h->outC.as<double>(), thresh ? h->outE.as<double>() : nullptr, nullptr, nullptr, h->exactSpread,
                          h->stream));
    MRC_TRY(s.down(smr, h->outC, szSmr));
    if (thresh) MRC_TRY(s.down(thresh, h->outE, szLines));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

static int bitalloc_host(mrc_handle* h, int64_t n_cases, int n_bands, int max_mant_bits, const int32_t* n_lines,
                         const double* budget, const double* smr, int32_t* bits, int32_t* bits_left, double* smr_after) {
    if (!h || !n_lines || !budget || !smr || !bits || !bits_left || n_cases < 0 || n_bands < 1 || n_bands > 64)
        return fail(h, MRC_ERR_INVALID, "mrc_bitalloc: bad argument (1 <= n_bands <= 64)");
    if (n_cases == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inAux, n_lines, (size_t)n_bands * sizeof(int32_t)));
    MRC_TRY(s.up(h->inAux2, budget, (size_t)n_cases * sizeof(double)));
    MRC_TRY(s.up(h->inL, smr, (size_t)n_cases * n_bands * sizeof(double)));
    const size_t szBits = (size_t)n_cases * n_bands * sizeof(int32_t), szLeft = (size_t)n_cases * sizeof(int32_t);
    const size_t szSmr = (size_t)n_cases * n_bands * sizeof(double);
    MRC_HIP(h, h->outC.reserve(szBits)); MRC_HIP(h, h->outF.reserve(szLeft));
    if (smr_after) MRC_HIP(h, h->outG.reserve(szSmr));
    MRC_HIP(h, launch_bitalloc_cases(n_cases, n_bands, max_mant_bits, h->inAux.as<int>(), h->inAux2.as<double>(),
                                     h->inL.as<double>(), h->outC.as<int>(), h->outF.as<int>(),
                                     smr_after ? h->outG.as<double>() : nullptr, h->stream));
    MRC_TRY(s.down(bits, h->outC, szBits));
    MRC_TRY(s.down(bits_left, h->outF, szLeft));
    if (smr_after) MRC_TRY(s.down(smr_after, h->outG, szSmr));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_bitalloc(mrc_handle* h, int64_t n_cases, int n_bands, int max_mant_bits, const int32_t* n_lines,
                 const double* budget, const double* smr, int32_t* bits, int32_t* bits_left) {
    return bitalloc_host(h, n_cases, n_bands, max_mant_bits, n_lines, budget, smr, bits, bits_left, nullptr);
}

int mrc_bitalloc_inplace(mrc_handle* h, int64_t n_cases, int n_bands, int max_mant_bits, const int32_t* n_lines,
                         const double* budget, double* smr, int32_t* bits, int32_t* bits_left) {
    return bitalloc_host(h, n_cases, n_bands, max_mant_bits, n_lines, budget, smr, bits, bits_left, smr);
}

int mrc_scale_factor(mrc_handle* h, int64_t n, int n_scale_bits, const double* v, const int32_t* n_mant_bits,
                     int32_t* scale) {
    if (!h || !v || !n_mant_bits || !scale || n < 0 || n_scale_bits < 1 || n_scale_bits > 4)
        return fail(h, MRC_ERR_INVALID, "mrc_scale_factor: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, v, (size_t)n * sizeof(double)));
    MRC_TRY(s.up(h->inAux, n_mant_bits, (size_t)n * sizeof(int32_t)));
    MRC_HIP(h, h->outC.reserve((size_t)n * sizeof(int32_t)));
    MRC_HIP(h, launch_scale_factor(n, n_scale_bits, h->inL.as<double>(), h->inAux.as<int>(), h->outC.as<int>(), h->stream));
    MRC_TRY(s.down(scale, h->outC, (size_t)n * sizeof(int32_t)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_mantissa(mrc_handle* h, int64_t n, int n_scale_bits, const double* x, const int32_t* scale,
                 const int32_t* n_mant_bits, int32_t* mant) {
    if (!h || !x || !scale || !n_mant_bits || !mant || n < 0 || n_scale_bits < 1 || n_scale_bits > 4)
        return fail(h, MRC_ERR_INVALID, "mrc_mantissa: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, x, (size_t)n * sizeof(double)));
    MRC_TRY(s.up(h->inAux, scale, (size_t)n * sizeof(int32_t)));
    MRC_TRY(s.up(h->inAux2, n_mant_bits, (size_t)n * sizeof(int32_t)));
    MRC_HIP(h, h->outC.reserve((size_t)n * sizeof(int32_t)));
    MRC_HIP(h, launch_mantissa(n, n_scale_bits, h->inL.as<double>(), h->inAux.as<int>(), h->inAux2.as<int>(),
                               h->outC.as<int>(), h->stream));
    MRC_TRY(s.down(mant, h->outC, (size_t)n * sizeof(int32_t)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_decode(mrc_handle* h, int64_t n, int a, int b, int n_streams, const int32_t* overall_scale,
               const int32_t* ms_switch, const int32_t* scale_factor, const int32_t* bit_alloc, const int32_t* mantissa,
               double* out) {
    if (!h || n < 0 || (n_streams != 1 && n_streams != 2) || !overall_scale || !scale_factor || !bit_alloc || !mantissa ||
        !out || (n_streams == 2 && !ms_switch))
        return fail(h, MRC_ERR_INVALID, "mrc_decode: bad argument");
    if (n == 0) return MRC_OK;
    const HostShape* hs;
    MRC_TRY(get_shape(h, a, b, &hs));
    const DevShape& S = hs->dev;
    MRC_HIP(h, hipSetDevice(h->device));
    std::vector<int64_t> offs((size_t)n);                   // (declared before `s`: alive until its drain)
    Staged s{h, h->stream};
    const size_t nOs = n_streams == 2 ? 4 : 1;
    const size_t szBand = (size_t)n * n_streams * S.nBands * sizeof(int32_t), szM = (size_t)n * n_streams * S.halfN * sizeof(int32_t);
    const size_t szOut = (size_t)n * n_streams * S.N * sizeof(double);
    MRC_TRY(s.up(h->inAux, overall_scale, (size_t)n * nOs * sizeof(int32_t)));
    MRC_TRY(s.up(h->inAux2, scale_factor, szBand));
    MRC_TRY(s.up(h->inAux3, bit_alloc, szBand));
    MRC_TRY(s.up(h->inL, mantissa, szM));
    if (n_streams == 2) MRC_TRY(s.up(h->inR, ms_switch, (size_t)n * S.nBands * sizeof(int32_t)));
    // block i, channel c -> out[(i * n_streams + c) * N]: one plane, per-channel base pointers and a common offset
    for (int64_t i = 0; i < n; ++i) offs[(size_t)i] = i * n_streams * (int64_t)S.N;
    MRC_TRY(s.up(h->outA, offs.data(), (size_t)n * sizeof(int64_t)));
    MRC_HIP(h, h->outG.reserve(szOut));
    MRC_HIP(h, hipMemsetAsync(h->outG.p, 0, szOut, h->stream));
    MRC_HIP(h, launch_decode(S, n, n_streams, h->inAux.as<int>(), n_streams == 2 ? h->inR.as<int>() : nullptr,
                             h->inAux2.as<int>(), h->inAux3.as<int>(), h->inL.as<int>(), h->outA.as<int64_t>(),
                             h->outG.as<double>(), h->outG.as<double>() + S.N, h->stream));
    MRC_TRY(s.down(out, h->outG, szOut));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_pcm16(mrc_handle* h, int64_t n, const double* x, int16_t* out) {
    if (!h || n < 0 || !x || !out) return fail(h, MRC_ERR_INVALID, "mrc_pcm16: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, x, (size_t)n * sizeof(double)));
    MRC_HIP(h, h->outA.reserve((size_t)n * sizeof(int16_t)));
    MRC_HIP(h, launch_pcm16(n, h->inL.as<double>(), h->outA.as<short>(), h->stream));
    MRC_TRY(s.down(out, h->outA, (size_t)n * sizeof(int16_t)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_huffman_gain(mrc_handle* h, int64_t n, int a, int b, int n_streams, const int32_t* bit_alloc,
                     const int32_t* mantissa, int32_t* huff_table, int32_t* bits_saved) {
    if (!h || !bit_alloc || !mantissa || !huff_table || !bits_saved || n < 0 || n_streams < 1 || n_streams > 4)
        return fail(h, MRC_ERR_INVALID, "mrc_huffman_gain: bad argument");
    if (n == 0) return MRC_OK;
    const HostShape* hs;
    MRC_TRY(get_shape(h, a, b, &hs));
    const DevShape& S = hs->dev;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    const size_t szBa = (size_t)n * n_streams * S.nBands * sizeof(int32_t);
    const size_t szM = (size_t)n * n_streams * S.halfN * sizeof(int32_t), szOut = (size_t)n * n_streams * sizeof(int32_t);
    MRC_TRY(s.up(h->inAux, bit_alloc, szBa));
    MRC_TRY(s.up(h->inL, mantissa, szM));
    MRC_HIP(h, h->outC.reserve(szOut)); MRC_HIP(h, h->outD.reserve(szOut));
    MRC_HIP(h, launch_huffman_gain(S, n, n_streams, h->inAux.as<int>(), h->inL.as<int>(), nullptr, h->outC.as<int>(),
                                   h->outD.as<int>(), nullptr, h->stream));
    MRC_TRY(s.down(huff_table, h->outC, szOut));
    MRC_TRY(s.down(bits_saved, h->outD, szOut));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

// every section behind the first has b0 == 1.0 exactly (scipy's tf2sos / zpk2sos put the gain into the first section)
static bool sos_unit_b0(const double* sos, int n_sections) {
    for (int s = 1; s < n_sections; ++s)
        if (sos[6 * s] != 1.0) return false;
    return true;
}

int mrc_transient_peaks_ex(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                           const void* streams, int sample_format, double* peaks) {
    if (!h || !sos || !streams || !peaks || n_hops < 0 || n_channels < 1 || n_sections < 1 || n_sections > 16 ||
        (sample_format != MRC_SAMPLES_F64 && sample_format != MRC_SAMPLES_PCM16))
        return fail(h, MRC_ERR_INVALID, "mrc_transient_peaks: bad argument (1 <= n_sections <= 16)");
    if (n_hops == 0) return MRC_OK;
    const int hop = h->cfg.n_mdct_lines, nShort = h->cfg.n_short;
    if (hop % nShort != 0) return fail(h, MRC_ERR_INVALID, "mrc_transient_peaks: n_mdct_lines must be a multiple of n_short");
    const int64_t chStride = (n_hops + 1) * (int64_t)hop;
    const size_t smp = sample_format == MRC_SAMPLES_PCM16 ? sizeof(int16_t) : sizeof(double);
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, streams, (size_t)n_channels * chStride * smp));
    MRC_TRY(s.up(h->inAux3, sos, (size_t)n_sections * 6 * sizeof(double)));
    const size_t outBytes = (size_t)n_hops * n_channels * (hop / nShort + 1) * sizeof(double);
    MRC_HIP(h, h->outG.reserve(outBytes));
    MRC_HIP(h, launch_transient_peaks(n_hops, n_channels, hop, nShort, n_sections, h->inAux3.as<double>(),
                                      sos_unit_b0(sos, n_sections), h->inL.p, sample_format, chStride, h->outG.as<double>(), h->stream));
    MRC_TRY(s.down(peaks, h->outG, outBytes));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_transient_peaks(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                        const double* streams, double* peaks) {
    return mrc_transient_peaks_ex(h, n_hops, n_channels, n_sections, sos, streams, MRC_SAMPLES_F64, peaks);
}

int mrc_dev_transient_peaks(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                            const void* streams, int sample_format, int64_t channel_stride, double* peaks, void* stream) {
    if (!h || !sos || !streams || !peaks || n_hops < 0 || n_channels < 1 || n_sections < 1 || n_sections > 16 ||
        (sample_format != MRC_SAMPLES_F64 && sample_format != MRC_SAMPLES_PCM16))
        return fail(h, MRC_ERR_INVALID, "mrc_dev_transient_peaks: bad argument (1 <= n_sections <= 16)");
    if (n_hops == 0) return MRC_OK;
    const int hop = h->cfg.n_mdct_lines, nShort = h->cfg.n_short;
    if (hop % nShort != 0) return fail(h, MRC_ERR_INVALID, "mrc_dev_transient_peaks: n_mdct_lines must be a multiple of n_short");
    if (channel_stride < (n_hops + 1) * (int64_t)hop) return fail(h, MRC_ERR_INVALID, "mrc_dev_transient_peaks: channel_stride too small");
    MRC_HIP(h, hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    // (the filter coefficients are 6 doubles per section: staged in the handle; the copy is ordered on `st`)
    MRC_HIP(h, h->inAux3.reserve((size_t)n_sections * 6 * sizeof(double)));
    MRC_HIP(h, hipMemcpyAsync(h->inAux3.p, sos, (size_t)n_sections * 6 * sizeof(double), hipMemcpyHostToDevice, st));
    MRC_HIP(h, hipStreamSynchronize(st));                    // `sos` may be a temporary of the caller
    MRC_HIP(h, launch_transient_peaks(n_hops, n_channels, hop, nShort, n_sections, h->inAux3.as<double>(),
                                      sos_unit_b0(sos, n_sections), streams, sample_format, channel_stride, peaks, st));
    return MRC_OK;
}

int mrc_stereo_masking_factor(mrc_handle* h, int64_t n, const double* mid_thresh, const double* side_thresh,
                              const double* z, double* out_mid, double* out_side) {
    if (!h || !mid_thresh || !side_thresh || !z || !out_mid || !out_side || n < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_stereo_masking_factor: bad argument");
    if (n == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    const size_t bytes = (size_t)n * sizeof(double);
    MRC_TRY(s.up(h->inL, mid_thresh, bytes));
    MRC_TRY(s.up(h->inR, side_thresh, bytes));
    MRC_TRY(s.up(h->inAux3, z, bytes));
    MRC_HIP(h, h->outG.reserve(bytes)); MRC_HIP(h, h->outE.reserve(bytes));
    MRC_HIP(h, launch_stereo_masking(n, h->inL.as<double>(), h->inR.as<double>(), h->inAux3.as<double>(),
                                     h->outG.as<double>(), h->outE.as<double>(), h->stream));
    MRC_TRY(s.down(out_mid, h->outG, bytes));
    MRC_TRY(s.down(out_side, h->outE, bytes));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_ms_switch(mrc_handle* h, int64_t n_blocks, int n_bands, const int32_t* n_lines, const double* lines_left,
                  const double* lines_right, int32_t* ms_switch) {
    if (!h || !n_lines || !lines_left || !lines_right || !ms_switch || n_blocks < 0 || n_bands < 1 ||
        n_bands > MRC_MAX_BANDS)
        return fail(h, MRC_ERR_INVALID, "mrc_ms_switch: bad argument");
    if (n_blocks == 0) return MRC_OK;
    std::vector<int> lo(n_bands), cnt(n_bands), plan;
    int total = 0;
    for (int i = 0; i < n_bands; ++i) {
        if (n_lines[i] < 0) return fail(h, MRC_ERR_INVALID, "mrc_ms_switch: negative band size");
        lo[i] = total; cnt[i] = n_lines[i]; total += n_lines[i];
    }
    int nLeaves = 0, nInternal = 0;
    ms_plan(lo, cnt, &plan, &nLeaves, &nInternal);
    if (nLeaves + nInternal > 64) return fail(h, MRC_ERR_INVALID, "mrc_ms_switch: band table too fine (more than 64 summation nodes)");
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};                                 // (plan, lo, cnt are declared before it: alive until its drain)
    MRC_TRY(s.up(h->inAux, plan.data(), plan.size() * sizeof(int)));
    MRC_TRY(s.up(h->inL, lines_left, (size_t)n_blocks * total * sizeof(double)));
    MRC_TRY(s.up(h->inR, lines_right, (size_t)n_blocks * total * sizeof(double)));
    MRC_HIP(h, h->outC.reserve((size_t)n_blocks * n_bands * sizeof(int32_t)));
    MRC_HIP(h, launch_ms_switch(n_blocks, n_bands, nLeaves, nInternal, h->inAux.as<int>(), h->inL.as<double>(),
                                h->inR.as<double>(), total, total, h->outC.as<int>(), h->stream));
    MRC_TRY(s.down(ms_switch, h->outC, (size_t)n_blocks * n_bands * sizeof(int32_t)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

}  // extern "C"
