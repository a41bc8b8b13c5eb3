// C ABI of libmrc_hip.so (include/mrc_hip.h): handle, shape cache, workspace, host<->device staging.
// No computation happens here and there is no CPU fallback: every entry point ends in a kernel launch
// of mrc_kernels.hip or fails.
#include "mrc_internal.hpp"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>

using namespace mrc;

namespace {

std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return (T*)p; }
};

}  // namespace

struct mrc_handle {
    mrc_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::pair<int, int>, HostShape> shapes;
    std::string error;
    // workspace of mrc_dev_encode
    DevBuf wsLines, wsScale, wsSmr, wsPeak;
    // staging of the host entry points
    DevBuf inL, inR, inAux, inAux2, inAux3, outA, outB, outC, outD, outE, outF, outG;
    bool timing = false;
    bool exactSpread = false;        // mrc_set_option(MRC_OPT_EXACT_SPREAD)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double stageMs[3] = {0, 0, 0};
};

namespace {

int fail(mrc_handle* h, int code, const std::string& msg) {
    if (h) h->error = msg; else g_create_error = msg;
    return code;
}

int hip_fail(mrc_handle* h, hipError_t e, const char* what) {
    return fail(h, MRC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define MRC_HIP(h, call)                                              \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) return hip_fail((h), e_, #call);        \
    } while (0)

int get_shape(mrc_handle* h, int a, int b, const HostShape** out) {
    auto key = std::make_pair(a, b);
    auto it = h->shapes.find(key);
    if (it == h->shapes.end()) {
        HostShape hs;
        std::string err;
        MRC_HIP(h, hipSetDevice(h->device));
        if (!build_shape(h->cfg, a, b, &hs, &err)) return fail(h, MRC_ERR_INVALID, err);
        it = h->shapes.emplace(key, std::move(hs)).first;
    }
    *out = &it->second;
    return MRC_OK;
}

hipStream_t pick_stream(mrc_handle* h, void* stream) { return stream ? (hipStream_t)stream : h->stream; }

}  // namespace

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
        if (rc != MRC_OK) { g_create_error = h->error; mrc_destroy(h); return rc; }
    }
    *out = h;
    return MRC_OK;
}

void mrc_destroy(mrc_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& kv : h->shapes) free_shape(&kv.second);
    for (DevBuf* b : {&h->wsLines, &h->wsScale, &h->wsSmr, &h->wsPeak, &h->inL, &h->inR, &h->inAux, &h->inAux2, &h->inAux3,
                      &h->outA, &h->outB, &h->outC, &h->outD, &h->outE, &h->outF, &h->outG})
        b->release();
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* mrc_last_error(const mrc_handle* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

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
    return fail(h, MRC_ERR_INVALID, "mrc_set_option: unknown option");
}

int mrc_get_stage_ms(mrc_handle* h, double* ms) {
    if (!h || !ms) return MRC_ERR_INVALID;
    for (int i = 0; i < 3; ++i) ms[i] = h->stageMs[i];
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
    MRC_HIP(h, launch_mdct(hs->dev, n_frames, ch_left, ch_right, frame_stride, offsets, true, lines, overall_scale,
                           pick_stream(h, stream)));
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
    MRC_HIP(h, launch_smr(hs->dev, n_frames, ch_left, ch_right, frame_stride, offsets, lines, overall_scale, smr,
                          thresh, nullptr, h->exactSpread, pick_stream(h, stream)));
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
    MRC_HIP(h, hipSetDevice(h->device));
    MRC_HIP(h, h->wsPeak.reserve(alloc_workspace_bytes(hs->dev, n_frames, joint)));
    MRC_HIP(h, launch_alloc_quant(hs->dev, n_frames, joint, lines, overall_scale, smr, reservoir_in, ms_switch,
                                  bit_alloc, scale_factor, mantissa, reservoir_out, h->wsPeak.as<double>(), false,
                                  pick_stream(h, stream)));
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

int mrc_dev_encode(mrc_handle* h, int a, int b, int64_t n_frames, const double* ch_left, const double* ch_right,
                   int64_t frame_stride, const int64_t* offsets, const int32_t* reservoir_in,
                   int32_t* overall_scale, int32_t* ms_switch, int32_t* bit_alloc, int32_t* scale_factor,
                   int32_t* mantissa, int32_t* reservoir_out, double* lines_out, void* stream) {
    if (!h || !ch_left || !overall_scale || !bit_alloc || !scale_factor || !mantissa || !reservoir_out ||
        (ch_right && !ms_switch) || n_frames < 0)
        return fail(h, MRC_ERR_INVALID, "mrc_dev_encode: bad argument");
    const HostShape* hs;
    int rc = get_shape(h, a, b, &hs);
    if (rc) return rc;
    const DevShape& S = hs->dev;
    const int joint = ch_right ? 1 : 0;
    const int nsig = joint ? 4 : 1;
    hipStream_t st = pick_stream(h, stream);
    MRC_HIP(h, hipSetDevice(h->device));
    double* lines = lines_out;
    if (!lines) {
        MRC_HIP(h, h->wsLines.reserve((size_t)n_frames * nsig * S.halfN * sizeof(double)));
        lines = h->wsLines.as<double>();
    }
    MRC_HIP(h, h->wsSmr.reserve((size_t)n_frames * nsig * S.nBands * sizeof(double)));
    MRC_HIP(h, h->wsPeak.reserve(alloc_workspace_bytes(S, n_frames, joint)));
    double* smr = h->wsSmr.as<double>();
    if (h->timing) MRC_HIP(h, hipEventRecord(h->ev[0], st));
    MRC_HIP(h, launch_mdct(S, n_frames, ch_left, ch_right, frame_stride, offsets, true, lines, overall_scale, st));
    if (h->timing) MRC_HIP(h, hipEventRecord(h->ev[1], st));
    MRC_HIP(h, launch_smr(S, n_frames, ch_left, ch_right, frame_stride, offsets, lines, overall_scale, smr, nullptr,
                          h->wsPeak.as<double>(), h->exactSpread, st));
    if (h->timing) MRC_HIP(h, hipEventRecord(h->ev[2], st));
    MRC_HIP(h, launch_alloc_quant(S, n_frames, joint, lines, overall_scale, smr, reservoir_in, ms_switch, bit_alloc,
                                  scale_factor, mantissa, reservoir_out, h->wsPeak.as<double>(), true, st));
    if (h->timing) {
        MRC_HIP(h, hipEventRecord(h->ev[3], st));
        MRC_HIP(h, hipEventSynchronize(h->ev[3]));
        for (int i = 0; i < 3; ++i) {
            float ms = 0.f;
            MRC_HIP(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
            h->stageMs[i] = ms;
        }
    }
    return MRC_OK;
}

// ---------------------------------------------------------------------------------------------- host API

namespace {

struct Staged {
    mrc_handle* h;
    hipStream_t st;
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

#define MRC_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != MRC_OK) return rc_; \
    } while (0)

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
    MRC_TRY(s.up(h->inL, left, inBytes));
    if (joint) MRC_TRY(s.up(h->inR, right, inBytes));
    if (reservoir_in) MRC_TRY(s.up(h->inAux, reservoir_in, (size_t)n * sizeof(int32_t)));
    const size_t szScale = (size_t)n * nsig * sizeof(int32_t), szSw = (size_t)n * S.nBands * sizeof(int32_t);
    const size_t szBand = (size_t)n * nstream * S.nBands * sizeof(int32_t);
    const size_t szMant = (size_t)n * nstream * S.halfN * sizeof(int32_t), szRes = (size_t)n * sizeof(int32_t);
    const size_t szLines = (size_t)n * nsig * S.halfN * sizeof(double);
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
    MRC_HIP(h, launch_mdct(S, n, h->inL.as<double>(), nullptr, S.N, nullptr, apply_window != 0, h->outG.as<double>(),
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
        MRC_HIP(h, launch_mdct(S, n, h->inL.as<double>(), nullptr, S.N, nullptr, true, h->outG.as<double>(),
                               h->outA.as<int>(), h->stream));
    }
    MRC_HIP(h, launch_smr(S, n, h->inL.as<double>(), nullptr, S.N, nullptr, h->outG.as<double>(), h->outA.as<int>(),
                          h->outC.as<double>(), thresh ? h->outE.as<double>() : nullptr, nullptr, h->exactSpread,
                          h->stream));
    MRC_TRY(s.down(smr, h->outC, szSmr));
    if (thresh) MRC_TRY(s.down(thresh, h->outE, szLines));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_bitalloc(mrc_handle* h, int64_t n_cases, int n_bands, int max_mant_bits, const int32_t* n_lines,
                 const double* budget, const double* smr, int32_t* bits, int32_t* bits_left) {
    if (!h || !n_lines || !budget || !smr || !bits || !bits_left || n_cases < 0 || n_bands < 1 || n_bands > 64)
        return fail(h, MRC_ERR_INVALID, "mrc_bitalloc: bad argument (1 <= n_bands <= 64)");
    if (n_cases == 0) return MRC_OK;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inAux, n_lines, (size_t)n_bands * sizeof(int32_t)));
    MRC_TRY(s.up(h->inAux2, budget, (size_t)n_cases * sizeof(double)));
    MRC_TRY(s.up(h->inL, smr, (size_t)n_cases * n_bands * sizeof(double)));
    const size_t szBits = (size_t)n_cases * n_bands * sizeof(int32_t), szLeft = (size_t)n_cases * sizeof(int32_t);
    MRC_HIP(h, h->outC.reserve(szBits)); MRC_HIP(h, h->outF.reserve(szLeft));
    MRC_HIP(h, launch_bitalloc_cases(n_cases, n_bands, max_mant_bits, h->inAux.as<int>(), h->inAux2.as<double>(),
                                     h->inL.as<double>(), h->outC.as<int>(), h->outF.as<int>(), h->stream));
    MRC_TRY(s.down(bits, h->outC, szBits));
    MRC_TRY(s.down(bits_left, h->outF, szLeft));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
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
    std::vector<int64_t> offs((size_t)n);
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

int mrc_transient_peaks(mrc_handle* h, int64_t n_hops, int n_channels, int n_sections, const double* sos,
                        const double* streams, double* peaks) {
    if (!h || !sos || !streams || !peaks || n_hops < 0 || n_channels < 1 || n_sections < 1 || n_sections > 16)
        return fail(h, MRC_ERR_INVALID, "mrc_transient_peaks: bad argument (1 <= n_sections <= 16)");
    if (n_hops == 0) return MRC_OK;
    const int hop = h->cfg.n_mdct_lines, nShort = h->cfg.n_short;
    if (hop % nShort != 0) return fail(h, MRC_ERR_INVALID, "mrc_transient_peaks: n_mdct_lines must be a multiple of n_short");
    const int64_t chStride = (n_hops + 1) * (int64_t)hop;
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inL, streams, (size_t)n_channels * chStride * sizeof(double)));
    MRC_TRY(s.up(h->inAux3, sos, (size_t)n_sections * 6 * sizeof(double)));
    const size_t outBytes = (size_t)n_hops * n_channels * (hop / nShort + 1) * sizeof(double);
    MRC_HIP(h, h->outG.reserve(outBytes));
    MRC_HIP(h, launch_transient_peaks(n_hops, n_channels, hop, nShort, n_sections, h->inAux3.as<double>(),
                                      h->inL.as<double>(), chStride, h->outG.as<double>(), h->stream));
    MRC_TRY(s.down(peaks, h->outG, outBytes));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
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
    std::vector<int> lo(n_bands), cnt(n_bands);
    int total = 0;
    for (int i = 0; i < n_bands; ++i) {
        if (n_lines[i] < 0) return fail(h, MRC_ERR_INVALID, "mrc_ms_switch: negative band size");
        lo[i] = total; cnt[i] = n_lines[i]; total += n_lines[i];
    }
    MRC_HIP(h, hipSetDevice(h->device));
    Staged s{h, h->stream};
    MRC_TRY(s.up(h->inAux, lo.data(), (size_t)n_bands * sizeof(int)));
    MRC_TRY(s.up(h->inAux2, cnt.data(), (size_t)n_bands * sizeof(int)));
    MRC_TRY(s.up(h->inL, lines_left, (size_t)n_blocks * total * sizeof(double)));
    MRC_TRY(s.up(h->inR, lines_right, (size_t)n_blocks * total * sizeof(double)));
    MRC_HIP(h, h->outC.reserve((size_t)n_blocks * n_bands * sizeof(int32_t)));
    MRC_HIP(h, launch_ms_switch(n_blocks, n_bands, total, h->inAux.as<int>(), h->inAux2.as<int>(), h->inL.as<double>(),
                                h->inR.as<double>(), h->outC.as<int>(), h->stream));
    MRC_TRY(s.down(ms_switch, h->outC, (size_t)n_blocks * n_bands * sizeof(int32_t)));
    MRC_HIP(h, hipStreamSynchronize(h->stream));    // lo/cnt (pageable host vectors) stay alive until here
    return MRC_OK;
}

}  // extern "C"
