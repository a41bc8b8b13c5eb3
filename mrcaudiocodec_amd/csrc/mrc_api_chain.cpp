// C ABI, chained stream encode (include/mrc_hip.h: mrc_encode_chained_stream_pcm16_pac, mrc_dev_encode_chained_pac):
// the encode direction of the reference's command line (pacfileThem.py:1159-1214, Close() 973-984, file header 586-613)
// for whole stereo streams in ONE call, block shapes in, `.pac` bytes out.
//
//   phase A   per block shape, ONE launch set over all blocks of all streams: windowed MDCT, overall scale, M/S switch,
//             SMRs, band peaks (the batch kernels) -- nothing here depends on the bit reservoir;
//   prep      per block: the bit allocation's grant events sorted (chain_prep_kernel);
//   phase B   one workgroup per stream walks its blocks in file order with the reservoir carried from block to block on
//             the device (chain_phase_b_kernel): bit allocation, scale factors, mantissas, Huffman pricing;
//   pack      per block shape plan / write kernels of the device packer around ONE prefix sum over all chunks in file
//             order, the file headers in front of every stream.
// No computation happens in this file.
#include "mrc_handle.hpp"

#include <cstring>
#include <vector>

using namespace mrc;

namespace {

struct SyncGuard {                    // whichever way we leave: nothing queued still reads the host vectors declared before it
    hipStream_t st;
    ~SyncGuard() { if (st) (void)hipStreamSynchronize(st); }
};

template <class T>
int upload(mrc_handle* h, DevBuf& buf, const std::vector<T>& v, hipStream_t st) {
    MRC_HIP(h, buf.reserve(v.empty() ? 1 : v.size() * sizeof(T)));
    if (!v.empty()) MRC_HIP(h, hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return MRC_OK;
}

}  // namespace

extern "C" {

int mrc_get_chain_ms(mrc_handle* h, double* ms) {
    if (!h || !ms) return MRC_ERR_INVALID;
    for (int i = 0; i < 4; ++i) ms[i] = h->chainMs[i];
    return MRC_OK;
}

int64_t mrc_chain_out_bound(mrc_handle* h, int64_t n_streams, const int64_t* block_start, const int32_t* block_a,
                            const int32_t* block_b, int with_flush, int with_headers) {
    if (!h || n_streams < 0 || !block_start || !block_a || !block_b) return MRC_ERR_INVALID;
    const int L = h->cfg.n_mdct_lines, Sh = h->cfg.n_short;
    // a stream has four block shapes: their bounds once, not one band table per block
    const int sa[4] = {L, L, Sh, Sh}, sb[4] = {L, Sh, Sh, L};
    int64_t shapeBound[4];
    for (int g = 0; g < 4; ++g) shapeBound[g] = mrc_pack_bound(&h->cfg, sa[g], sb[g], 2, 1);
    int64_t total = 0;
    for (int64_t i = block_start[0]; i < block_start[n_streams]; ++i) {
        const int g = (block_a[i] == L ? 0 : 2) + ((block_a[i] == L) == (block_b[i] == L) ? 0 : 1);
        int64_t bnd = shapeBound[g];
        if (block_a[i] != sa[g] || block_b[i] != sb[g]) bnd = mrc_pack_bound(&h->cfg, block_a[i], block_b[i], 2, 1);   // (refused later)
        if (bnd < 0) return MRC_ERR_INVALID;
        total += bnd;
    }
    if (with_flush) total += n_streams * mrc_pack_bound(&h->cfg, L, L, 2, 0);
    if (with_headers) total += n_streams * 128;
    return total;
}

}  // extern "C"

namespace {

// One SLAB of a chained encode: all of the streams [0, n_streams) handed over, every buffer sized for exactly these blocks
// (the entry points below cut a call into slabs).
int chained_core(mrc_handle* h, int64_t n_streams, const void* pcm_left, const void* pcm_right,
                 int sample_format, int64_t stream_stride, const int64_t* block_start, const int64_t* block_offset,
                 const int32_t* block_a, const int32_t* block_b, const int32_t* reservoir_in,
                 int use_huffman, int with_flush, const uint32_t* num_samples, uint8_t* out, int64_t out_cap,
                 int64_t* stream_byte_offset, int64_t* item_byte_offset, int32_t* reservoir_out,
                 int32_t* reservoir_trace, int64_t* total_bytes, void* stream) {
    if (!h || n_streams < 0 || !pcm_left || !pcm_right || stream_stride <= 0 || !block_start || !block_offset ||
        !block_a || !block_b || !out || out_cap < 0 || !stream_byte_offset || !total_bytes ||
        (sample_format != MRC_SAMPLES_F64 && sample_format != MRC_SAMPLES_PCM16))
        return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: bad argument");
    const size_t sampleBytes = sample_format == MRC_SAMPLES_PCM16 ? sizeof(int16_t) : sizeof(double);
    *total_bytes = 0;
    stream_byte_offset[0] = 0;
    if (n_streams == 0) return MRC_OK;
    const mrc_config& cfg = h->cfg;
    const int L = cfg.n_mdct_lines, Sh = cfg.n_short;
    const int64_t b0 = block_start[0], nB = block_start[n_streams] - b0;
    if (nB < n_streams) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: every stream needs at least one block");
    // ---- the block shapes of the reference's block switching (pacfileThem.py:1192-1210); group 4: Close()'s blocks
    const int shapeA[kChainGroups] = {L, L, Sh, Sh, L}, shapeB[kChainGroups] = {L, Sh, Sh, L, L};
    const int nGroups = with_flush ? kChainGroups : kChainGroups - 1;
    const HostShape* hs[kChainGroups] = {};
    for (int g = 0; g < nGroups; ++g) {
        MRC_TRY(get_shape(h, shapeA[g], shapeB[g], &hs[g]));
        const DevShape& S = hs[g]->dev;
        const int nTot = (g == 4 ? 1 : 2) * S.nBands;
        if (nTot > 64 || S.maxMantBits < 2 || S.maxMantBits > 16 || (S.halfN & 3))
            return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: shape outside what the chained back end covers "
                                            "(<= 32 bands, 2..16 mantissa bits, lines a multiple of 4)");
    }
    // ---- the schedule, pass 1: validate, sort the blocks into their shape groups (the offsets phase A needs).  The rest of
    // the schedule (items in file order, chunk maps, headers) is only needed by the serial scan and the packer: it is built
    // and uploaded in pass 2, AFTER phase A's launches are queued, so the device works while the host prepares it.
    const int64_t nItems = nB + (with_flush ? 2 * n_streams : 0);
    const int64_t nChunks = 2 * nB + (with_flush ? 2 * n_streams : 0);
    std::vector<uint8_t> groupOf((size_t)nB);
    std::vector<int64_t> offs[kChainGroups];
    std::vector<long long> tailOff((size_t)n_streams);
    offs[0].reserve((size_t)nB);
    for (int64_t s = 0; s < n_streams; ++s) {
        const int64_t i0 = block_start[s], i1 = block_start[s + 1];
        if (i1 <= i0) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: every stream needs at least one block");
        for (int64_t i = i0; i < i1; ++i) {
            const int a = block_a[i], b = block_b[i];
            int g = -1;
            for (int q = 0; q < 4; ++q) if (a == shapeA[q] && b == shapeB[q]) { g = q; break; }   // (L == Sh: group 0)
            if (g < 0) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: block shape is not one of (L,L), (L,S), (S,S), (S,L)");
            const int64_t off = block_offset[i];
            if (off < 0 || off + a + b > stream_stride)
                return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: block reaches outside its stream");
            if (offs[g].size() >= (size_t)1 << 28) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: too many blocks of one shape");
            groupOf[(size_t)(i - b0)] = (uint8_t)g;
            offs[g].push_back(s * stream_stride + off);
        }
        if (with_flush) {
            if (block_b[i1 - 1] != L)
                return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: a stream must end with a long block (the reference's "
                                                "Close() assumes it, pacfileThem.py:973-984)");
            tailOff[(size_t)s] = block_offset[i1 - 1] + block_a[i1 - 1];
        }
    }

    MRC_HIP(h, hipSetDevice(h->device));
    hipStream_t st = pick_stream(h, stream);
    ChainBufs& C = h->chain;
    for (auto& e : C.evT) if (!e) MRC_HIP(h, hipEventCreate(&e));
    std::vector<long long> pos(item_byte_offset ? (size_t)nChunks + 1 : 0), streamPos((size_t)n_streams);
    std::vector<int32_t> resOut((size_t)n_streams);
    // (filled in pass 2; declared here: the guard below outlives every host buffer a queued copy may still read)
    std::vector<int32_t> items, chunkStream, resIn;
    std::vector<long long> itemStart, firstChunk, itemChunk, chunkMap[kChainGroups];
    std::vector<uint8_t> hdr;
    long long total = 0;
    int bad = 0;
    SyncGuard guard{st};
    MRC_HIP(h, hipEventRecord(C.evT[0], st));
    if (with_flush) {
        // the tail offsets ride in the offsets buffer of group 4 (its blocks are laid out explicitly, stride 2 L)
        MRC_TRY(upload(h, C.g[4].offsets, tailOff, st));
        MRC_HIP(h, C.flushPcm.reserve((size_t)n_streams * 2 * 2 * L * sampleBytes));
        MRC_HIP(h, launch_chain_flush_gather(n_streams, L, pcm_left, pcm_right, sample_format, stream_stride,
                                             C.g[4].offsets.as<long long>(), C.flushPcm.p, st));
    }
    // ---- phase A + prep, per block shape
    ChainGroupDev desc[kChainGroups];
    std::memset(desc, 0, sizeof(desc));
    int64_t count[kChainGroups] = {};
    for (int g = 0; g < nGroups; ++g) {
        const DevShape& S = hs[g]->dev;
        const int joint = g == 4 ? 0 : 1, nsig = joint ? 4 : 1, nstream = joint ? 2 : 1;
        const int64_t m = g == 4 ? 2 * n_streams : (int64_t)offs[g].size();
        count[g] = m;
        ChainGroupBufs& B = C.g[g];
        const int nTot = nstream * S.nBands, nEv = (int)chain_events_per_block(S, joint);
        if (m > 0) {
            if (g != 4) MRC_TRY(upload(h, B.offsets, offs[g], st));
            MRC_HIP(h, B.lines.reserve((size_t)m * nsig * S.halfN * sizeof(double)));
            MRC_HIP(h, B.oscale.reserve((size_t)m * nsig * sizeof(int32_t)));
            MRC_HIP(h, B.smr.reserve((size_t)m * nsig * S.nBands * sizeof(double)));
            MRC_HIP(h, B.peak.reserve((size_t)m * nsig * S.nBands * sizeof(double)));
            MRC_HIP(h, B.ms.reserve((size_t)m * S.nBands * sizeof(int32_t)));
            MRC_HIP(h, B.ev.reserve((size_t)m * nEv * sizeof(unsigned)));
            MRC_HIP(h, B.pre.reserve((size_t)m * (nEv + 1) * sizeof(unsigned)));
            MRC_HIP(h, B.bitAlloc.reserve((size_t)m * nTot * sizeof(int32_t)));
            MRC_HIP(h, B.scaleFactor.reserve((size_t)m * nTot * sizeof(int32_t)));
            MRC_HIP(h, B.mant.reserve((size_t)m * nstream * S.halfN * sizeof(uint16_t)));
            MRC_HIP(h, B.table.reserve((size_t)m * nstream * sizeof(int32_t)));
            if (g == 4)
                MRC_TRY(encode_phase_a(h, S, m, C.flushPcm.p, nullptr, sample_format, 2 * (int64_t)L, nullptr, B.lines.as<double>(),
                                       B.oscale.as<int32_t>(), nullptr, B.smr.as<double>(), B.peak.as<double>(), st, false));
            else
                MRC_TRY(encode_phase_a(h, S, m, pcm_left, pcm_right, sample_format, 0, B.offsets.as<int64_t>(),
                                       B.lines.as<double>(), B.oscale.as<int32_t>(), B.ms.as<int32_t>(), B.smr.as<double>(),
                                       B.peak.as<double>(), st, false));
            MRC_HIP(h, launch_chain_prep(S, joint, m, B.smr.as<double>(), joint ? B.ms.as<int32_t>() : nullptr,
                                         B.ev.as<unsigned>(), B.pre.as<unsigned>(),
                                         h->chainForceFallback ? 1 : 0, st));
        }
        ChainGroupDev& D = desc[g];
        D.joint = joint; D.nb = S.nBands; D.nTot = nTot; D.M = S.halfN; D.K = S.maxMantBits - 1; D.nEv = nEv;
        D.nScaleBits = S.nScaleBits; D.nstream = nstream;
        D.maxN = 0;
        for (int v : hs[g]->bandN) if (v > D.maxN) D.maxN = v;
        D.budgetMono = S.budgetMono; D.budgetJointPre = S.budgetJointPre; D.blkswA = S.blkswA; D.blkswB = S.blkswB;
        D.bandOfLine = S.bandOfLine; D.bandN = S.bandN;
        D.lines = B.lines.as<double>(); D.peak = B.peak.as<double>(); D.oscale = B.oscale.as<int32_t>();
        D.ms = B.ms.as<int32_t>(); D.ev = B.ev.as<unsigned>();
        D.pre = B.pre.as<unsigned>();
        D.bitAlloc = B.bitAlloc.as<int32_t>(); D.scaleFactor = B.scaleFactor.as<int32_t>();
        D.mant = B.mant.as<unsigned short>(); D.table = B.table.as<int32_t>();
    }
    // ---- the schedule, pass 2 (the device is busy with phase A): items (group << 28 | index inside the group) per stream in
    // file order, the chunk of every item, the stream of every chunk, the chunks of every group
    items.resize((size_t)nItems);
    itemStart.resize((size_t)n_streams + 1); firstChunk.resize((size_t)n_streams);
    itemChunk.resize((size_t)nItems + 1);
    chunkStream.resize((size_t)nChunks);
    resIn.assign((size_t)n_streams, 0);
    for (int g = 0; g < 4; ++g) chunkMap[g].resize(2 * offs[g].size());
    if (with_flush) chunkMap[4].resize((size_t)2 * n_streams);
    {
        int64_t it = 0, ch = 0;
        size_t idx[kChainGroups] = {};
        for (int64_t s = 0; s < n_streams; ++s) {
            itemStart[(size_t)s] = it;
            firstChunk[(size_t)s] = ch;
            for (int64_t i = block_start[s]; i < block_start[s + 1]; ++i) {
                const int g = groupOf[(size_t)(i - b0)];
                const size_t k = idx[g]++;
                items[(size_t)it] = (int32_t)((unsigned)g << 28 | (unsigned)k);
                chunkMap[g][2 * k] = ch; chunkMap[g][2 * k + 1] = ch + 1;
                itemChunk[(size_t)it] = ch;
                chunkStream[(size_t)ch] = chunkStream[(size_t)ch + 1] = (int32_t)s;
                ch += 2; ++it;
            }
            if (with_flush)
                for (int c = 0; c < 2; ++c) {                      // codec.Encode: channel after channel
                    items[(size_t)it] = (int32_t)(4u << 28 | (unsigned)(2 * s + c));
                    chunkMap[4][(size_t)(2 * s + c)] = ch;
                    itemChunk[(size_t)it] = ch;
                    chunkStream[(size_t)ch] = (int32_t)s;
                    ++ch; ++it;
                }
            if (reservoir_in) resIn[(size_t)s] = reservoir_in[s];
        }
        itemStart[(size_t)n_streams] = it;
        itemChunk[(size_t)nItems] = ch;
    }
    // ---- file headers (pacfileThem.py:586-613)
    int hdrLen = 0;
    if (num_samples) {
        // one header built by mrc_pac_header; the streams differ only in the sample count (bytes 10..13, little endian, with
        // the reference's padding rule, pacfileThem.py:595-597: padded when it ALREADY is a multiple of nMDCTLines)
        uint8_t one[256];
        int64_t len = 0;
        if (mrc_pac_header(&cfg, 2, num_samples[0], one, sizeof(one), &len) != MRC_OK || len < 14)
            return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: mrc_pac_header failed");
        hdrLen = (int)len;
        hdr.resize((size_t)n_streams * len);
        for (int64_t s = 0; s < n_streams; ++s) {
            uint8_t* dst = hdr.data() + s * len;
            std::memcpy(dst, one, (size_t)len);
            uint32_t ns = num_samples[s];
            if (ns % (uint32_t)cfg.n_mdct_lines == 0) ns += (uint32_t)cfg.n_mdct_lines;
            for (int q = 0; q < 4; ++q) dst[10 + q] = (uint8_t)(ns >> (8 * q));
        }
    }
    MRC_TRY(upload(h, C.items, items, st));
    MRC_TRY(upload(h, C.itemStart, itemStart, st));
    MRC_TRY(upload(h, C.reservoir, resIn, st));
    MRC_TRY(upload(h, C.chunkStream, chunkStream, st));
    MRC_TRY(upload(h, C.hdr, hdr, st));
    MRC_TRY(upload(h, C.firstChunk, firstChunk, st));
    for (int g = 0; g < nGroups; ++g)
        if (count[g] > 0) MRC_TRY(upload(h, C.g[g].chunkMap, chunkMap[g], st));
    if (reservoir_trace) MRC_HIP(h, C.resTrace.reserve((size_t)nItems * sizeof(int32_t)));
    MRC_HIP(h, C.groupDesc.reserve(sizeof(desc)));
    MRC_HIP(h, hipMemcpyAsync(C.groupDesc.p, desc, sizeof(desc), hipMemcpyHostToDevice, st));
    MRC_HIP(h, hipEventRecord(C.evT[1], st));
    // ---- phase B: the serial scan per stream
    MRC_HIP(h, launch_chain_phase_b(n_streams, C.groupDesc.as<ChainGroupDev>(), C.items.as<int>(), C.itemStart.as<long long>(),
                                    C.reservoir.as<int>(), reservoir_trace ? C.resTrace.as<int>() : nullptr,
                                    use_huffman ? 1 : 0, h->chainThreads, st));
    MRC_HIP(h, hipEventRecord(C.evT[2], st));
    if (h->sensOn)                                       // MRC_OPT_SENSITIVITY: the scan's decisions, group by group
        for (int g = 0; g < nGroups; ++g) {
            ChainGroupBufs& B = C.g[g];
            const int joint = g == 4 ? 0 : 1;
            MRC_HIP(h, launch_sensitivity(hs[g]->dev, count[g], joint, B.lines.as<double>(), B.oscale.as<int32_t>(),
                                          B.smr.as<double>(), B.peak.as<double>(), joint ? B.ms.as<int32_t>() : nullptr,
                                          B.bitAlloc.as<int32_t>(), B.scaleFactor.as<int32_t>(),
                                          h->sens.as<unsigned long long>(), nullptr, st));
        }
    // ---- pack: plan per shape, ONE prefix sum over the chunks in file order, write per shape
    static const PackTables tables = [] { PackTables t; pack_tables(&t); return t; }();
    MRC_HIP(h, C.packWs.reserve(pack_workspace_bytes(nChunks)));
    const PackWs W = pack_ws_views(C.packWs.p, nChunks);
    MRC_HIP(h, hipMemsetAsync(W.errorFlag, 0, sizeof(int), st));
    PackParams P[kChainGroups];
    for (int g = 0; g < nGroups; ++g) {
        const int joint = g == 4 ? 0 : 1;
        P[g].nch = joint ? 2 : 1; P[g].joint = joint; P[g].useHuffman = use_huffman ? 1 : 0;
        P[g].nScaleBits = cfg.n_scale_bits; P[g].nMantSizeBits = cfg.n_mant_size_bits;
        P[g].blkBitsA = cfg.blksw_bits_a; P[g].blkBitsB = cfg.blksw_bits_b;
        P[g].bitA = (unsigned)(1 - shapeA[g] / cfg.n_mdct_lines); P[g].bitB = (unsigned)(1 - shapeB[g] / cfg.n_mdct_lines);
        if (!count[g]) continue;
        ChainGroupBufs& B = C.g[g];
        const int64_t nBlk = count[g];                              // (a mono item is a one-channel block)
        MRC_HIP(h, launch_pack_plan(hs[g]->dev, P[g], tables, nBlk, B.bitAlloc.as<int>(), B.mant.p, MRC_MANTISSA_I16,
                                    B.table.as<int>(), B.table.as<int>(), nullptr, W, B.chunkMap.as<long long>(),
                                    all_bands_non_empty(*hs[g]), st));
    }
    MRC_HIP(h, launch_pack_scan(nChunks, 0, W, nullptr, num_samples ? C.chunkStream.as<int>() : nullptr, hdrLen, st));
    for (int g = 0; g < nGroups; ++g) {
        if (!count[g]) continue;
        ChainGroupBufs& B = C.g[g];
        const int bound = (int)(mrc_pack_bound(&cfg, shapeA[g], shapeB[g], 1, P[g].joint) - 4);
        MRC_HIP(h, launch_pack_write(hs[g]->dev, P[g], tables, count[g], B.oscale.as<int>(), P[g].joint ? B.ms.as<int>() : nullptr,
                                     B.scaleFactor.as<int>(), B.bitAlloc.as<int>(), B.mant.p, MRC_MANTISSA_I16,
                                     B.table.as<int>(), W, B.chunkMap.as<long long>(), out, (long long)out_cap, bound,
                                     all_bands_non_empty(*hs[g]), st));
    }
    // the file headers (num_samples given), and the start of every stream's bytes
    MRC_HIP(h, C.streamPos.reserve((size_t)n_streams * sizeof(long long)));
    MRC_HIP(h, launch_chain_headers(n_streams, hdrLen, C.hdr.as<unsigned char>(), C.firstChunk.as<long long>(), W.pos, out,
                                    (long long)out_cap, C.streamPos.as<long long>(), st));
    MRC_HIP(h, hipEventRecord(C.evT[3], st));
    // ---- results: stream starts (the position of every chunk only if the caller asked for them), total, error flag,
    // reservoirs
    if (item_byte_offset)
        MRC_HIP(h, hipMemcpyAsync(pos.data(), W.pos, pos.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipMemcpyAsync(streamPos.data(), C.streamPos.p, streamPos.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipMemcpyAsync(&total, W.total, sizeof(total), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipMemcpyAsync(&bad, W.errorFlag, sizeof(bad), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipMemcpyAsync(resOut.data(), C.reservoir.p, resOut.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (reservoir_trace)
        MRC_HIP(h, hipMemcpyAsync(reservoir_trace, C.resTrace.p, (size_t)nItems * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MRC_HIP(h, hipStreamSynchronize(st));
    for (int i = 0; i < 3; ++i) {
        float ms = 0.f;
        MRC_HIP(h, hipEventElapsedTime(&ms, C.evT[i], C.evT[i + 1]));
        h->chainMs[i] = ms;
    }
    {
        float ms = 0.f;
        MRC_HIP(h, hipEventElapsedTime(&ms, C.evT[0], C.evT[3]));
        h->chainMs[3] = ms;
    }
    *total_bytes = total;
    for (int64_t s = 0; s < n_streams; ++s) stream_byte_offset[s] = streamPos[(size_t)s];
    stream_byte_offset[n_streams] = total;
    if (item_byte_offset) {
        for (int64_t i = 0; i < nItems; ++i) item_byte_offset[i] = pos[(size_t)itemChunk[(size_t)i]];
        item_byte_offset[nItems] = total;
    }
    if (reservoir_out) std::memcpy(reservoir_out, resOut.data(), resOut.size() * sizeof(int32_t));
    if (bad & 3) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: internal error (table id / chunk size out of range)");
    if (total > out_cap || (bad & 4)) return fail(h, MRC_ERR_NOMEM, "mrc_encode_chained: out_cap too small (see total_bytes)");
    return MRC_OK;
}

// ---- slabs (round 4).  Phase A keeps ~45 KB of device memory per joint long block (the MDCT lines of four signals, SMRs,
// events, outputs) and the worst-case output bound is 13 KB per block: a call over a 2^18-hop file would hold 18 GB.  A call is
// therefore cut into SLABS of at most h->chainSlabBlocks blocks, each a chained_core of its own whose buffers are reused by
// the next: whole streams while they fit (their files stay contiguous in the output), a stream longer than a slab alone in
// consecutive TIME slabs -- the reservoir goes from slab to slab as it goes from block to block (codecThem.py:274,503), the
// header travels with the first slab, Close()'s blocks with the last.  `sink` receives each slab's bytes.
struct Slab { int64_t s0, ns; int64_t i0, i1; bool first, last, timeSlab; };

std::vector<Slab> plan_slabs(int64_t n_streams, const int64_t* block_start, int64_t cap) {
    std::vector<Slab> v;
    int64_t s = 0;
    while (s < n_streams) {
        const int64_t nb = block_start[s + 1] - block_start[s];
        if (nb > cap) {                                  // one long stream: time slabs
            for (int64_t i = block_start[s]; i < block_start[s + 1]; i += cap) {
                const int64_t e = std::min<int64_t>(i + cap, block_start[s + 1]);
                v.push_back({s, 1, i, e, i == block_start[s], e == block_start[s + 1], true});
            }
            ++s;
            continue;
        }
        int64_t e = s, blocks = 0;
        while (e < n_streams && block_start[e + 1] - block_start[e] <= cap && blocks + (block_start[e + 1] - block_start[e]) <= cap) {
            blocks += block_start[e + 1] - block_start[e];
            ++e;
        }
        v.push_back({s, e - s, block_start[s], block_start[e], true, true, false});
        s = e;
    }
    return v;
}

// sink(slab bytes are at `buf` on the device, n of them, they belong at byte `at` of the call's output) -> status
template <class Sink>
int chained_slabs(mrc_handle* h, int64_t n_streams, const void* pcm_left, const void* pcm_right, int sample_format,
                  int64_t stream_stride, const int64_t* block_start, const int64_t* block_offset, const int32_t* block_a,
                  const int32_t* block_b, const int32_t* reservoir_in, int use_huffman, int with_flush,
                  const uint32_t* num_samples, int64_t out_cap, int64_t* stream_byte_offset, int64_t* item_byte_offset,
                  int32_t* reservoir_out, int32_t* reservoir_trace, int64_t* total_bytes, void* stream,
                  uint8_t* direct_out /* device buffer of out_cap bytes to write into in place, or null: C.out per slab */,
                  Sink sink) {
    if (!h || n_streams < 0 || !block_start || !stream_byte_offset || !total_bytes)
        return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: bad argument");
    *total_bytes = 0;
    stream_byte_offset[0] = 0;
    if (n_streams == 0) return MRC_OK;
    for (int64_t s = 0; s < n_streams; ++s)
        if (block_start[s + 1] <= block_start[s]) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: every stream needs at least one block");
    const size_t sampleBytes = sample_format == MRC_SAMPLES_PCM16 ? sizeof(int16_t) : sizeof(double);
    const std::vector<Slab> slabs = plan_slabs(n_streams, block_start, h->chainSlabBlocks > 0 ? h->chainSlabBlocks : (int64_t)1 << 40);
    ChainBufs& C = h->chain;
    C.lastTotal = -1;
    int64_t written = 0, itemBase = 0;
    bool overflow = false;
    double ms[4] = {0, 0, 0, 0};
    std::vector<int64_t> sOff, iOff;
    int32_t carry = 0;
    for (const Slab& sl : slabs) {
        const char* pl = (const char*)pcm_left + (size_t)sl.s0 * stream_stride * sampleBytes;
        const char* pr = (const char*)pcm_right + (size_t)sl.s0 * stream_stride * sampleBytes;
        const int64_t bs2[2] = {sl.i0, sl.i1};
        const int64_t* bs = sl.timeSlab ? bs2 : block_start + sl.s0;
        const int flush = with_flush && sl.last;
        const uint32_t* nsamp = (num_samples && sl.first) ? num_samples + sl.s0 : nullptr;
        const int32_t* resIn = sl.timeSlab ? (sl.first ? (reservoir_in ? reservoir_in + sl.s0 : nullptr) : &carry)
                                           : (reservoir_in ? reservoir_in + sl.s0 : nullptr);
        const int64_t nItems = (sl.i1 - sl.i0) + (flush ? 2 * sl.ns : 0);
        sOff.assign((size_t)sl.ns + 1, 0);
        if (item_byte_offset) iOff.assign((size_t)nItems + 1, 0);
        int64_t slabTotal = 0;
        uint8_t* dst;
        int64_t cap;
        if (direct_out && !overflow) { dst = direct_out + written; cap = out_cap - written; }
        else {
            const int64_t bound = mrc_chain_out_bound(h, sl.ns, bs, block_a, block_b, flush, nsamp != nullptr);
            if (bound < 0) return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: block shape out of range");
            MRC_HIP(h, hipSetDevice(h->device));
            MRC_HIP(h, C.out.reserve((size_t)bound + 1));
            dst = C.out.as<uint8_t>(); cap = bound;
        }
        int32_t resOutSlab[1] = {0};
        int rc = chained_core(h, sl.ns, pl, pr, sample_format, stream_stride, bs, block_offset, block_a, block_b, resIn,
                              use_huffman, flush, nsamp, dst, cap, sOff.data(), item_byte_offset ? iOff.data() : nullptr,
                              sl.timeSlab ? resOutSlab : (reservoir_out ? reservoir_out + sl.s0 : nullptr),
                              reservoir_trace ? reservoir_trace + itemBase : nullptr, &slabTotal, stream);
        if (rc == MRC_ERR_NOMEM && direct_out) overflow = true;          // the caller's buffer is full: sizes only from here on
        else if (rc != MRC_OK) return rc;
        for (int i = 0; i < 4; ++i) ms[i] += h->chainMs[i];
        if (sl.timeSlab) {
            carry = resOutSlab[0];
            if (sl.first) stream_byte_offset[sl.s0] = written + sOff[0];
            if (sl.last) { stream_byte_offset[sl.s0 + 1] = written + slabTotal; if (reservoir_out) reservoir_out[sl.s0] = carry; }
        } else {
            for (int64_t s = 0; s <= sl.ns; ++s) stream_byte_offset[sl.s0 + s] = written + sOff[(size_t)s];
        }
        if (item_byte_offset)
            for (int64_t i = 0; i <= nItems; ++i) item_byte_offset[itemBase + i] = written + iOff[(size_t)i];
        if (!direct_out && !overflow) {
            if (written + slabTotal > out_cap) overflow = true;
            else MRC_TRY(sink(dst, slabTotal, written));
        }
        written += slabTotal;
        itemBase += nItems;
    }
    for (int i = 0; i < 4; ++i) h->chainMs[i] = ms[i];
    *total_bytes = written;
    stream_byte_offset[n_streams] = written;
    if (slabs.size() == 1 && !direct_out) C.lastTotal = written;         // (one slab: its bytes are all in C.out, mrc_chain_fetch_output)
    if (overflow || written > out_cap) return fail(h, MRC_ERR_NOMEM, "mrc_encode_chained: out_cap too small (see total_bytes)");
    return MRC_OK;
}

}  // namespace

extern "C" {

int mrc_dev_encode_chained_pac(mrc_handle* h, int64_t n_streams, const void* pcm_left, const void* pcm_right,
                               int sample_format, int64_t stream_stride, const int64_t* block_start, const int64_t* block_offset,
                               const int32_t* block_a, const int32_t* block_b, const int32_t* reservoir_in,
                               int use_huffman, int with_flush, const uint32_t* num_samples, uint8_t* out, int64_t out_cap,
                               int64_t* stream_byte_offset, int64_t* item_byte_offset, int32_t* reservoir_out,
                               int32_t* reservoir_trace, int64_t* total_bytes, void* stream) {
    if (!h || !pcm_left || !pcm_right || stream_stride <= 0 || !block_offset || !block_a || !block_b || !out || out_cap < 0 ||
        (sample_format != MRC_SAMPLES_F64 && sample_format != MRC_SAMPLES_PCM16))
        return fail(h, MRC_ERR_INVALID, "mrc_encode_chained: bad argument");
    return chained_slabs(h, n_streams, pcm_left, pcm_right, sample_format, stream_stride, block_start, block_offset, block_a,
                         block_b, reservoir_in, use_huffman, with_flush, num_samples, out_cap, stream_byte_offset,
                         item_byte_offset, reservoir_out, reservoir_trace, total_bytes, stream, out,
                         [](uint8_t*, int64_t, int64_t) { return (int)MRC_OK; });
}

int mrc_encode_chained_stream_pac(mrc_handle* h, int64_t n_streams, const void* pcm_left, const void* pcm_right,
                                  int sample_format, int64_t stream_stride, const int64_t* block_start,
                                  const int64_t* block_offset, const int32_t* block_a, const int32_t* block_b,
                                  const int32_t* reservoir_in, int use_huffman, int with_flush, const uint32_t* num_samples,
                                  uint8_t* out, int64_t out_cap, int64_t* stream_byte_offset, int64_t* item_byte_offset,
                                  int32_t* reservoir_out, int32_t* reservoir_trace, int64_t* total_bytes) {
    if (!h || n_streams < 0 || !pcm_left || !pcm_right || stream_stride <= 0 || !out || !total_bytes || !block_start ||
        (sample_format != MRC_SAMPLES_F64 && sample_format != MRC_SAMPLES_PCM16))
        return fail(h, MRC_ERR_INVALID, "mrc_encode_chained_stream_pac: bad argument");
    MRC_HIP(h, hipSetDevice(h->device));
    ChainBufs& C = h->chain;
    const size_t pcmBytes = (size_t)n_streams * stream_stride * (sample_format == MRC_SAMPLES_PCM16 ? sizeof(int16_t) : sizeof(double));
    MRC_HIP(h, C.pcmL.reserve(pcmBytes ? pcmBytes : 1));
    MRC_HIP(h, C.pcmR.reserve(pcmBytes ? pcmBytes : 1));
    SyncGuard guard{h->stream};
    if (pcmBytes) {
        MRC_HIP(h, hipMemcpyAsync(C.pcmL.p, pcm_left, pcmBytes, hipMemcpyHostToDevice, h->stream));
        MRC_HIP(h, hipMemcpyAsync(C.pcmR.p, pcm_right, pcmBytes, hipMemcpyHostToDevice, h->stream));
    }
    // every slab packs into the handle's device buffer (sized for the slab's worst case) and its bytes are copied behind the
    // previous slab's in the caller's buffer, which only has to hold what the streams really pack to
    hipStream_t st = h->stream;
    mrc_handle* hh = h;
    int rc = chained_slabs(h, n_streams, C.pcmL.p, C.pcmR.p, sample_format, stream_stride, block_start, block_offset, block_a,
                           block_b, reservoir_in, use_huffman, with_flush, num_samples, out_cap, stream_byte_offset,
                           item_byte_offset, reservoir_out, reservoir_trace, total_bytes, h->stream, nullptr,
                           [out, st, hh](uint8_t* buf, int64_t n, int64_t at) {
                               if (n) MRC_HIP(hh, hipMemcpyAsync(out + at, buf, (size_t)n, hipMemcpyDeviceToHost, st));
                               MRC_HIP(hh, hipStreamSynchronize(st));      // (the next slab reuses the buffer)
                               return (int)MRC_OK;
                           });
    if (rc == MRC_ERR_NOMEM)
        return fail(h, MRC_ERR_NOMEM, "mrc_encode_chained_stream_pac: out_cap too small (see total_bytes; mrc_chain_fetch_output)");
    return rc;
}

int mrc_chain_fetch_output(mrc_handle* h, uint8_t* out, int64_t out_cap, int64_t* total_bytes) {
    if (!h || !out || !total_bytes) return fail(h, MRC_ERR_INVALID, "mrc_chain_fetch_output: bad argument");
    ChainBufs& C = h->chain;
    if (C.lastTotal < 0) return fail(h, MRC_ERR_INVALID, "mrc_chain_fetch_output: no output of a chained call is held");
    *total_bytes = C.lastTotal;
    if (C.lastTotal > out_cap) return fail(h, MRC_ERR_NOMEM, "mrc_chain_fetch_output: out_cap too small (see total_bytes)");
    MRC_HIP(h, hipSetDevice(h->device));
    if (C.lastTotal) MRC_HIP(h, hipMemcpyAsync(out, C.out.p, (size_t)C.lastTotal, hipMemcpyDeviceToHost, h->stream));
    MRC_HIP(h, hipStreamSynchronize(h->stream));
    return MRC_OK;
}

int mrc_encode_chained_stream_pcm16_pac(mrc_handle* h, int64_t n_streams, const int16_t* pcm_left, const int16_t* pcm_right,
                                        int64_t stream_stride, const int64_t* block_start, const int64_t* block_offset,
                                        const int32_t* block_a, const int32_t* block_b, const int32_t* reservoir_in,
                                        int use_huffman, int with_flush, const uint32_t* num_samples, uint8_t* out,
                                        int64_t out_cap, int64_t* stream_byte_offset, int64_t* item_byte_offset,
                                        int32_t* reservoir_out, int32_t* reservoir_trace, int64_t* total_bytes) {
    return mrc_encode_chained_stream_pac(h, n_streams, pcm_left, pcm_right, MRC_SAMPLES_PCM16, stream_stride, block_start,
                                         block_offset, block_a, block_b, reservoir_in, use_huffman, with_flush, num_samples,
                                         out, out_cap, stream_byte_offset, item_byte_offset, reservoir_out, reservoir_trace,
                                         total_bytes);
}

}  // extern "C"
