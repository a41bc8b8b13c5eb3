// The handle behind the C ABI and the helpers every translation unit of the host glue shares (mrc_api.cpp: the per-block
// and pipelined entry points; mrc_api_chain.cpp: the chained stream encode).  Not part of the ABI.
#pragma once
#include "mrc_internal.hpp"

#include <map>
#include <string>
#include <utility>

namespace mrc {

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

// page-locked host staging of the small-batch host entry points (one copy each way instead of one per array)
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// intermediate results of one encode call (lines, SMRs, band peaks); one set per stream that encodes concurrently
struct Workspace {
    DevBuf lines, smr, peak;
    void release() { lines.release(); smr.release(); peak.release(); }
};

// The pipelined host entry point runs THREE streams -- one that only copies in, one that only launches kernels, one that
// only copies out -- over a ring of chunk buffers (lanes), ordered by events: measured on the MI355X box, page-locked
// copies reach 42-48 GB/s each way with ONE stream per direction and drop to 19-25 GB/s with three streams that each
// copy both ways (tools/pcie_rates.py), which is what one-stream-per-chunk pipelining amounts to.
struct Lane {
    DevBuf pcmL, pcmR, resIn, oScale, ms, ba, sf, mant, resOut;
    DevBuf pacBytes, pacOffs, pacTable, pacSaved;                // mrc_encode_stream_pcm16_pac: the chunk's packed form
    long long* pacTotal = nullptr;                               // page-locked: the chunk's byte count, read by the host
    hipEvent_t evIn = nullptr, evK = nullptr, evOut = nullptr;   // chunk copied in / encoded / copied out
    void release() {
        for (DevBuf* b : {&pcmL, &pcmR, &resIn, &oScale, &ms, &ba, &sf, &mant, &resOut, &pacBytes, &pacOffs, &pacTable, &pacSaved})
            b->release();
        if (pacTotal) (void)hipHostFree(pacTotal);
        pacTotal = nullptr;
        for (hipEvent_t* e : {&evIn, &evK, &evOut}) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    }
};
constexpr int kLanes = 4;          // chunk buffers in flight (mrc_encode_stream_pcm16_pac reads sizes two chunks behind)
constexpr int kKernelEvents = 6;   // boundaries of: mdct | smr | band_stats | bitalloc | quantize
constexpr int kChainGroups = 5;    // chained encode: the four joint block shapes + Close()'s non-joint long block

// Device state of one block-shape group of the chained encode (mrc_api_chain.cpp): the reservoir-free results (phase A),
// the sorted grant events of the bit allocation, and the outputs of the serial scan (phase B)
struct ChainGroupBufs {
    DevBuf offsets, lines, oscale, smr, peak, ms;                // phase A
    DevBuf ev, pre;                                              // prepared for phase B
    DevBuf bitAlloc, scaleFactor, mant, table, chunkMap;         // phase B outputs, packer inputs
    void release() {
        for (DevBuf* b : {&offsets, &lines, &oscale, &smr, &peak, &ms, &ev, &pre, &bitAlloc, &scaleFactor, &mant, &table,
                          &chunkMap})
            b->release();
    }
};
struct ChainBufs {
    ChainGroupBufs g[kChainGroups];
    DevBuf pcmL, pcmR, flushPcm, items, itemStart, reservoir, groupDesc, packWs, out, hdr, chunkStream, resTrace, firstChunk,
        streamPos;
    hipEvent_t evT[4] = {};          // phase timing: start | phase A done | phase B done | packed
    int64_t lastTotal = -1;          // bytes the last mrc_encode_chained_stream_pac left in `out` (-1: none) -- mrc_chain_fetch_output
    void release() {
        for (auto& x : g) x.release();
        for (DevBuf* b : {&pcmL, &pcmR, &flushPcm, &items, &itemStart, &reservoir, &groupDesc, &packWs, &out, &hdr,
                          &chunkStream, &resTrace, &firstChunk, &streamPos})
            b->release();
        for (auto& e : evT) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    }
};

}  // namespace mrc

struct mrc_handle {
    mrc_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::pair<int, int>, mrc::HostShape> shapes;
    std::string error;
    mrc::Workspace ws;               // workspace of mrc_dev_encode* (calls on one handle are serialised)
    mrc::Lane lanes[mrc::kLanes];    // mrc_encode_stream_pcm16: chunk buffers ...
    hipStream_t stIn = nullptr, stOut = nullptr;   // ... and its copy-in / copy-out streams; the kernels of all chunks run
    mrc::Workspace wsPipe;           //     on `stream`, one after the other: one workspace.  (No third stream of its own:
                                     //     the runtime multiplexes streams onto 4 hardware queues by default -- with the
                                     //     null stream and `stream` that is exactly four; a fifth would share a queue with
                                     //     one of the others and serialise with it: 10 000 instead of 19 000 Msamples/s)
    // staging of the host entry points
    mrc::DevBuf inL, inR, inAux, inAux2, inAux3, outA, outB, outC, outD, outE, outF, outG;
    mrc::PinnedBuf pinIn, pinOut;    // ... of calls with few blocks (the per-block seam): one H2D, one D2H
    mrc::DevBuf packWs;              // mrc_dev_pack_blocks: chunk sizes / positions / (table ids)
    int64_t packLastChunks = 0, packLastCap = 0;   // ... of the most recent call (mrc_dev_pack_status)
    mrc::ChainBufs chain;            // mrc_encode_chained_*: see mrc_api_chain.cpp
    double chainMs[4] = {0, 0, 0, 0};   // last chained encode: phase A, phase B, pack, whole call (host clock)
    bool timing = false;
    bool exactSpread = false;        // mrc_set_option(MRC_OPT_EXACT_SPREAD)
    bool smrAllBands = false;        // mrc_set_option(MRC_OPT_SMR_ALL_BANDS)
    int chainThreads = 0;            // mrc_set_option(MRC_OPT_CHAIN_THREADS): workgroup size of the serial scan, 0 = by stream count
    int64_t chainSlabBlocks = 131072; // mrc_set_option(MRC_OPT_CHAIN_SLAB_BLOCKS): blocks per slab of a chained encode (~7.5 GB of device memory)
    bool chainForceFallback = false; // mrc_set_option(MRC_OPT_CHAIN_FORCE_REPAIR): tests of chain_prep_kernel's repair pass
    bool sensOn = false;             // mrc_set_option(MRC_OPT_SENSITIVITY): count decisions near a rounding edge ...
    mrc::DevBuf sens;                // ... here: MRC_SENS_COUNT counters (uint64), mrc_get_sensitivity
    hipEvent_t ev[mrc::kKernelEvents] = {};
    double stageMs[3] = {0, 0, 0};
    double kernelMs[5] = {0, 0, 0, 0, 0};
};

namespace mrc {

// error text of the last failed mrc_create (no handle exists yet)
std::string& create_error();

inline int fail(mrc_handle* h, int code, const std::string& msg) {
    if (h) h->error = msg; else create_error() = msg;
    return code;
}
inline int hip_fail(mrc_handle* h, hipError_t e, const char* what) {
    return fail(h, MRC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define MRC_HIP(h, call)                                              \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) return mrc::hip_fail((h), e_, #call);   \
    } while (0)
#define MRC_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != MRC_OK) return rc_; \
    } while (0)

// every entry point that launches comes through here first: the launches, the tables and the caller's pointers all
// belong to the handle's device, whatever the calling thread's current device was
int get_shape(mrc_handle* h, int a, int b, const HostShape** out);
inline bool all_bands_non_empty(const HostShape& hs) {
    for (int n : hs.bandN) if (n <= 0) return false;
    return true;
}
inline hipStream_t pick_stream(mrc_handle* h, void* stream) { return stream ? (hipStream_t)stream : h->stream; }

// phase A of the per-block path (windowed MDCT + overall scale -> [M/S switch] -> SMRs and per-band peaks), defined in
// mrc_api.cpp beside encode_core, which it is the first half of
int encode_phase_a(mrc_handle* h, const DevShape& S, int64_t n, const void* chL, const void* chR, int fmt, int64_t stride,
                   const int64_t* offsets, double* lines, int32_t* oscale, int32_t* msSwitch, double* smr, double* peak,
                   hipStream_t st, bool timing);

}  // namespace mrc
