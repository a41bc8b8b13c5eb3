"""
Host-side `.pac` writing for the encode path: the Huffman table choice and the bit packing run in C++
(libmrc_hip.so: mrc_pack_blocks / mrc_pack_joint_blocks / mrc_pac_header, no GPU needed for them), fed
with the dense arrays the GPU path returns.  Mirrors the encode half of the reference's file layer:

    PACFile.WriteFileHeader      pacfileThem.py:586-619   -> header()
    PACFile.WriteDataBlock       pacfileThem.py:622-790   -> pack_blocks()
    PACFile.JointWriteDataBlock  pacfileThem.py:793-972   -> pack_joint_blocks()
    the CLI's encode loop + Close pacfileThem.py:1159-1214, 973-984 -> encode_stereo_stream()

Block shapes are an input here; mrcaudiocodec_amd/transient.py derives them from the audio like the reference's
transient detector, and mrcaudiocodec_amd/cli.py strings WAV ingest, detector and this writer together.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, MrcConfig, MrcError

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def make_config(sample_rate=48000, n_mdct_lines=1024, n_short=128, n_scale_bits=4, n_mant_size_bits=4,
                target_bits_per_sample=2.86, blksw_bits_a=1, blksw_bits_b=1):
    cfg = MrcConfig()
    lib.mrc_default_config(C.byref(cfg))
    cfg.sample_rate, cfg.n_mdct_lines, cfg.n_short = int(sample_rate), int(n_mdct_lines), int(n_short)
    cfg.n_scale_bits, cfg.n_mant_size_bits = int(n_scale_bits), int(n_mant_size_bits)
    cfg.target_bits_per_sample = float(target_bits_per_sample)
    cfg.blksw_bits_a, cfg.blksw_bits_b = int(blksw_bits_a), int(blksw_bits_b)
    return cfg


def set_threads(n):
    """Host threads the C++ packer / parser uses for a batch of blocks (process-wide; default: the CPUs the process
    may run on, at most 16, or $MRC_PACK_THREADS)."""
    _check(lib.mrc_pack_set_threads(int(n)), "mrc_pack_set_threads")


def get_threads():
    return int(lib.mrc_pack_get_threads())


def _check(rc, what):
    if rc != 0:
        raise MrcError("%s failed (%d)" % (what, rc))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def band_table(cfg, a, b):
    n = C.c_int32()
    buf = np.zeros(_lib.MRC_MAX_BANDS, dtype=np.int32)
    _check(lib.mrc_band_table(C.byref(cfg), int(a), int(b), C.byref(n), buf.ctypes.data_as(_i32p)), "mrc_band_table")
    return buf[:n.value].copy()


def header(cfg, n_channels, num_samples):
    out = np.zeros(256, dtype=np.uint8)
    n = C.c_int64()
    _check(lib.mrc_pac_header(C.byref(cfg), int(n_channels), int(num_samples), out.ctypes.data_as(_u8p), out.size,
                              C.byref(n)), "mrc_pac_header")
    return out[:n.value].tobytes()


_BOUNDS = {}                    # mrc_pack_bound per (shape, channels, codec parameters): it walks a band table per call
# mrc_pack_blocks_ex with untyped pointer arguments (the typed prototype of _lib makes every array go through a cast)
_pack_blocks_ex = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                              C.c_void_p, C.c_void_p)(("mrc_pack_blocks_ex", lib))


def _pack(cfg, a, b, joint, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa, use_huffman, huff_table):
    """mrc_pack_blocks_ex: the mantissa plane goes in as it is -- int32, or the uint16 codes the PCM16 / mantissa16
    encode paths deliver (an int16 view of them is taken as uint16)."""
    sf, ba, osc = _i32(scale_factor), _i32(bit_alloc), _i32(overall_scale)
    n, nch = sf.shape[0], sf.shape[1]
    m = np.asarray(mantissa)
    if m.dtype == np.int16:
        m = m.view(np.uint16)
    if m.dtype == np.uint16:
        m, fmt = np.ascontiguousarray(m), 1
    else:
        m, fmt = _i32(m), 0
    sw = None if ms_switch is None else _i32(ms_switch)
    ht = None if huff_table is None else _i32(huff_table).reshape(n, nch)
    key = (int(a), int(b), nch, int(joint), cfg.n_mdct_lines, cfg.n_scale_bits, cfg.n_mant_size_bits, cfg.blksw_bits_a,
           cfg.blksw_bits_b, cfg.sample_rate)
    bound = _BOUNDS.get(key)
    if bound is None:
        bound = lib.mrc_pack_bound(C.byref(cfg), int(a), int(b), nch, int(joint))
        if bound < 0:
            raise MrcError("mrc_pack_bound failed (%d)" % bound)
        _BOUNDS[key] = bound
    out = np.empty(max(1, n * bound), dtype=np.uint8)          # every byte up to offs[n] is written by the packer
    offs = np.empty(n + 1, dtype=np.int64)
    table = np.empty((n, nch), dtype=np.int32) if ht is None else None
    saved = np.empty((n, nch), dtype=np.int32) if ht is None else None
    ptr = lambda arr: None if arr is None else C.c_void_p(arr.ctypes.data)      # (cheaper than data_as with a pointer type)
    rc = _pack_blocks_ex(C.byref(cfg), n, nch, int(a), int(b), int(joint), int(bool(use_huffman)), ptr(ht), ptr(osc), ptr(sw),
                         ptr(sf), ptr(ba), ptr(m), fmt, ptr(out), out.size, ptr(offs), ptr(table), ptr(saved))
    _check(rc, "mrc_pack_blocks_ex")
    return out[:offs[n]], offs, (ht if ht is not None else table), saved


def pack_blocks(cfg, a, b, overall_scale, scale_factor, bit_alloc, mantissa, use_huffman=True, huff_table=None):
    """WriteDataBlock for n blocks of nch independent channels.  overall_scale [n][nch], scale_factor /
    bit_alloc [n][nch][nBands], mantissa [n][nch][N/2] dense (int32 or uint16).  -> (bytes array, block offsets [n+1],
    huffTable [n][nch], bits_saved [n][nch]).  huff_table [n][nch] given (e.g. by Handle.dev_huffman_gain): the
    host skips the pricing of the four tables and bits_saved is None."""
    return _pack(cfg, a, b, 0, overall_scale, None, scale_factor, bit_alloc, mantissa, use_huffman, huff_table)


def pack_joint_blocks(cfg, a, b, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa, use_huffman=True,
                      huff_table=None):
    """JointWriteDataBlock for n blocks.  overall_scale [n][4], ms_switch [n][nBands], others [n][2][...];
    huff_table [n][2] as in pack_blocks."""
    return _pack(cfg, a, b, 1, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa, use_huffman, huff_table)


def encode_stereo_stream(handle, stream, shapes, use_huffman=True, num_samples=None):
    """The encode half of the reference CLI for ONE stereo stream [2][samples] (float64 signed fractions or int16 PCM
    codes) that starts with the zero prior hop and a given block-shape sequence [(offset, a, b)]: header, one joint block
    per shape with the bit reservoir chained through the Huffman savings (codecThem.py:274,503), then Close()'s flush
    block through the non-joint writer (pacfileThem.py:973-984).  One library call (mrc_encode_chained_stream_pac): the
    reservoir-free 95 % of the work runs as a batch over all blocks, the rest as a serial scan on the device, the
    packer on the device too.  Returns the .pac bytes."""
    stream = np.asarray(stream)
    return encode_stereo_streams(handle, stream[None], [shapes], use_huffman,
                                 None if num_samples is None else [num_samples])[0]


def encode_stereo_streams(handle, streams, shapes, use_huffman=True, num_samples=None):
    """encode_stereo_stream for MANY stereo streams at once: streams [nStreams][2][samples], shapes[s] = block-shape
    sequence of stream s, num_samples[s] (optional) = the header's sample count (default: the samples the blocks
    cover).  Returns a list of .pac byte strings, each identical to what the reference's driver writes for that
    stream alone."""
    L = handle.cfg.n_mdct_lines
    streams = np.asarray(streams)
    if streams.dtype != np.int16:
        streams = streams.astype(np.float64, copy=False)
    nS = streams.shape[0]
    if streams.ndim != 3 or streams.shape[1] != 2 or len(shapes) != nS:
        raise ValueError("streams [nStreams][2][samples] and one shape list per stream expected")
    for sh in shapes:
        if not len(sh) or sh[-1][2] != L:
            raise ValueError("every stream must end with a long block (the reference's Close() assumes it)")
    if num_samples is None:
        num_samples = [sum(int(b) for (_, _, b) in sh) for sh in shapes]       # the CLI writes the WAV's count
    r = handle.encode_chained_pac(streams[:, 0], streams[:, 1], shapes, use_huffman=use_huffman, with_flush=True,
                                  num_samples=num_samples)
    data, offs = r["bytes"], r["stream_offset"]
    return [data[offs[s]:offs[s + 1]].tobytes() for s in range(nS)]


def encode_stereo_stream_per_block(handle, stream, shapes, use_huffman=True, num_samples=None):
    """The block-at-a-time form of encode_stereo_stream, as the reference's loop runs it (and as this package ran it
    before the chained call existed): one mrc_encode_joint per block, the reservoir carried on the host, C++ packer.
    Kept as the cross-check of the chained path (same bytes) and as the 'before' of its speed-up."""
    c = handle.cfg
    cfg = make_config(c.sample_rate, c.n_mdct_lines, c.n_short, c.n_scale_bits, c.n_mant_size_bits,
                      c.target_bits_per_sample, c.blksw_bits_a, c.blksw_bits_b)
    L = c.n_mdct_lines
    if shapes[-1][2] != L:
        raise ValueError("the stream must end with a long block (the reference's Close() assumes it)")
    stream = np.asarray(stream, dtype=np.float64)
    out = [header(cfg, 2, sum(b for (_, _, b) in shapes) if num_samples is None else num_samples)]   # the CLI writes the WAV's count
    reservoir = 0
    for (off, a, b) in shapes:
        r = handle.encode_joint(stream[0, off:off + a + b][None, :], stream[1, off:off + a + b][None, :], a, b, [reservoir])
        data, _, _, saved = pack_joint_blocks(cfg, a, b, r["overall_scale"], r["ms_switch"], r["scale_factor"],
                                              r["bit_alloc"], r["mantissa"], use_huffman)
        reservoir = int(r["reservoir_out"][0]) + int(saved.sum())
        out.append(data.tobytes())
    off, a, b = shapes[-1]
    a = b
    for ch in range(2):                                         # Close: codec.Encode, channel after channel
        blk = np.concatenate([stream[ch, off + shapes[-1][1]:off + shapes[-1][1] + b], np.zeros(L)])[None, :]
        r = handle.encode_mono(blk, a, L, [reservoir])
        data, _, _, saved = pack_blocks(cfg, a, L, r["overall_scale"][:, None], r["scale_factor"][:, None, :],
                                        r["bit_alloc"][:, None, :], r["mantissa"][:, None, :], use_huffman)
        reservoir = int(r["reservoir_out"][0]) + int(saved.sum())
        out.append(data.tobytes())
    return b"".join(out)


# ------------------------------------------------------------------ decode side ("next" row f-4)
def read_header(buf):
    """pacfileThem.py:130-158 -> (cfg with the file's parameters, nChannels, numSamples, offset of the first chunk)."""
    raw = np.frombuffer(buf, dtype=np.uint8)
    cfg = make_config()
    nch, ns, off = C.c_int32(), C.c_uint32(), C.c_int64()
    _check(lib.mrc_pac_read_header(raw.ctypes.data_as(_u8p), raw.size, C.byref(cfg), C.byref(nch), C.byref(ns),
                                   C.byref(off)), "mrc_pac_read_header")
    return cfg, nch.value, ns.value, off.value


def scan_chunks(buf, data_offset):
    raw = np.frombuffer(buf, dtype=np.uint8)
    n = lib.mrc_pac_scan_chunks(raw.ctypes.data_as(_u8p), raw.size, int(data_offset), None, 0)
    if n < 0:
        raise MrcError("truncated .pac chunk")
    offs = np.zeros(max(n, 1), dtype=np.int64)
    lib.mrc_pac_scan_chunks(raw.ctypes.data_as(_u8p), raw.size, int(data_offset), offs.ctypes.data_as(_i64p), n)
    return offs[:n]


def unpack_blocks(cfg, buf, chunk_offsets, n_channels, joint):
    """The parsing half of (Joint)ReadDataBlock for the blocks whose chunks start at chunk_offsets
    [n * n_channels] -> dict of fixed-stride arrays (see mrc_unpack_blocks)."""
    raw = np.frombuffer(buf, dtype=np.uint8)
    offs = np.ascontiguousarray(chunk_offsets, dtype=np.int64)
    n = offs.size // n_channels
    L, B = cfg.n_mdct_lines, _lib.MRC_MAX_BANDS
    out = dict(a=np.zeros(n, np.int32), b=np.zeros(n, np.int32), huff_table=np.zeros((n, n_channels), np.int32),
               overall_scale=np.zeros((n, 4 if joint else n_channels), np.int32), ms_switch=np.zeros((n, B), np.int32),
               scale_factor=np.zeros((n, n_channels, B), np.int32), bit_alloc=np.zeros((n, n_channels, B), np.int32),
               mantissa=np.zeros((n, n_channels, L), np.int32))
    p = lambda k: out[k].ctypes.data_as(_i32p)
    _check(lib.mrc_unpack_blocks(C.byref(cfg), n, n_channels, int(bool(joint)), raw.ctypes.data_as(_u8p), raw.size,
                                 offs.ctypes.data_as(_i64p), p("a"), p("b"), p("huff_table"), p("overall_scale"),
                                 p("ms_switch"), p("scale_factor"), p("bit_alloc"), p("mantissa")), "mrc_unpack_blocks")
    return out


def decode_pac(handle, buf):
    """Decode a whole `.pac` byte string on the GPU of `handle` (which must have been created with the file's
    parameters): C++ chunk parser on the host, then per block shape one launch of the fused dequantise / M-S /
    IMDCT / window / overlap-add kernel into a device-resident output stream.  Returns (nChannels, float64
    [nCh][samples]): the concatenation of what the reference's successive (Joint)ReadDataBlock calls return --
    out[:, :L] is the half-block delay of the MDCT, the signal follows.  A stereo file is joint blocks followed by
    the two non-joint chunks Close() wrote (pacfileThem.py:973-984)."""
    import torch
    cfg, nch, _, off = read_header(buf)
    c = handle.cfg
    for k in ("sample_rate", "n_mdct_lines", "n_scale_bits", "n_mant_size_bits"):
        if getattr(cfg, k) != getattr(c, k):
            raise ValueError("handle was created with %s = %d, the file has %d" % (k, getattr(c, k), getattr(cfg, k)))
    cfg.n_short, cfg.blksw_bits_a, cfg.blksw_bits_b = c.n_short, c.blksw_bits_a, c.blksw_bits_b
    chunks = scan_chunks(buf, off)
    if nch not in (1, 2) or len(chunks) % nch:
        raise ValueError("unsupported channel count / chunk count")
    n_blocks = len(chunks) // nch
    groups = []                                              # (joint, parsed blocks) in file order
    if nch == 2 and n_blocks > 1:
        groups.append((True, unpack_blocks(cfg, buf, chunks[:2 * (n_blocks - 1)], 2, True)))
        groups.append((False, unpack_blocks(cfg, buf, chunks[2 * (n_blocks - 1):], 2, False)))
    else:
        groups.append((False, unpack_blocks(cfg, buf, chunks, nch, False)))
    a_all = np.concatenate([g["a"] for _, g in groups])
    b_all = np.concatenate([g["b"] for _, g in groups])
    starts = np.concatenate([[0], np.cumsum(a_all)[:-1]]).astype(np.int64)      # block i adds to [start, start + a + b)
    total = int(starts[-1] + a_all[-1] + b_all[-1]) if n_blocks else 0
    dev = torch.device("cuda", c.device_id)
    out = torch.zeros((nch, total), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    keep = []                                                # device inputs stay alive until the sync
    base = 0
    for joint, g in groups:
        n = len(g["a"])
        shapes = sorted(set(zip(g["a"].tolist(), g["b"].tolist())))
        for (a, b) in shapes:
            idx = np.nonzero((g["a"] == a) & (g["b"] == b))[0]
            nb, half = len(handle.bands(a, b)), (a + b) // 2
            offs = torch.from_numpy(starts[base + idx]).to(dev)
            up = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            if joint:
                args = (up(g["overall_scale"][idx]), up(g["ms_switch"][idx, :nb]), up(g["scale_factor"][idx][:, :, :nb]),
                        up(g["bit_alloc"][idx][:, :, :nb]), up(g["mantissa"][idx][:, :, :half]))
                handle.dev_decode(a, b, len(idx), 2, args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(),
                                  args[3].data_ptr(), args[4].data_ptr(), offs.data_ptr(), out[0].data_ptr(),
                                  out[1].data_ptr(), stream)
                keep.append((args, offs))
            else:
                for ch in range(nch):                        # independent channels: one mono launch per channel
                    args = (up(g["overall_scale"][idx, ch]), up(g["scale_factor"][idx, ch, :nb][:, None, :]),
                            up(g["bit_alloc"][idx, ch, :nb][:, None, :]), up(g["mantissa"][idx, ch, :half][:, None, :]))
                    handle.dev_decode(a, b, len(idx), 1, args[0].data_ptr(), None, args[1].data_ptr(),
                                      args[2].data_ptr(), args[3].data_ptr(), offs.data_ptr(), out[ch].data_ptr(), None,
                                      stream)
                    keep.append((args, offs))
        base += n
    torch.cuda.synchronize(dev)
    return nch, out


def decode_pac_pcm16(handle, buf):
    """decode_pac + the 16-bit PCM codes of pcmfile.py:163-172, with the first block (the MDCT's half-block delay)
    dropped as the reference's decode loop does (pacfileThem.py:1176-1179).  -> int16 [nCh][samples] (host)."""
    import torch
    nch, x = decode_pac(handle, buf)
    L = handle.cfg.n_mdct_lines
    x = x[:, L:].contiguous()
    pcm = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    handle.dev_pcm16(x.numel(), x.data_ptr(), pcm.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    return pcm.cpu().numpy()
