"""
Encode direction of the reference's command line (`python pacfileThem.py in.wav`, pacfileThem.py:1064-1231)
on the MI355X path:  python -m mrcaudiocodec_amd.cli in.wav out.pac [--no-huffman] [--device N]

WAV ingest (pcmfile.py:34-102: 16-bit PCM, int16 code c -> sign(c) 2|c|/65535), transient detection with
one hop of look-ahead (pacfileThem.py:1025-1056, 1182-1214), joint-stereo blocks with the bit reservoir
chained through the Huffman savings, Close()'s flush block, `.pac` framing -- kernels on the GPU, Huffman
and bit packing in C++ on the host.  Like the reference: stereo input only, and the last hop of the file is
analysed but never encoded.

Decode direction ("next" row f-4):  python -m mrcaudiocodec_amd.cli -d in.pac out.wav
C++ chunk parser on the host, dequantise / M-S / IMDCT / window / overlap-add and the 16-bit PCM codes on the GPU,
WAV header of pcmfile.py:141-153.  The first decoded block (the MDCT's half-block delay) is dropped as in the
reference's loop; everything after it is written, header sample count = what was decoded.
"""
import argparse
from struct import pack, unpack

import numpy as np

from . import Handle, pacfile, transient


def read_wav_pcm(path, hop=1024):
    """-> (sample_rate, n_channels, num_samples, int16 [nCh][nHops*hop]): the file's own codes, the last hop zero padded.
    The float map of pcmfile.py:91-100 (x = sign(c) 2|c| / 65535, -32768 -> 0.0) is applied on the device, on load."""
    with open(path, "rb") as fp:
        head = fp.read(12)
        if head[0:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file")
        while True:
            tag = fp.read(4)
            if len(tag) < 4:
                raise ValueError("no 'fmt ' chunk")
            if tag == b"fmt ":
                break
        (_, fmt, n_ch, rate, _, _, bits) = unpack("<LHHLLHH", fp.read(20))
        if fmt != 1 or bits != 16:
            raise ValueError("only 16-bit PCM WAV files are supported")
        while True:
            tag = fp.read(4)
            if len(tag) < 4:
                raise ValueError("no 'data' chunk")
            if tag == b"data":
                break
        num_samples = unpack("<L", fp.read(4))[0] // (n_ch * 2)
        raw = fp.read(num_samples * n_ch * 2)
    codes = np.frombuffer(raw, dtype="<i2")
    codes = codes[:(len(codes) // n_ch) * n_ch].reshape(-1, n_ch).T
    n_hops = -(-codes.shape[1] // hop)
    pcm = np.zeros((n_ch, n_hops * hop), np.int16)
    pcm[:, :codes.shape[1]] = codes
    return rate, n_ch, num_samples, pcm


def read_wav(path, hop=1024):
    """-> (sample_rate, n_channels, num_samples, float64 [nCh][nHops*hop]) as pcmfile.py:91-100 hands the samples on."""
    rate, n_ch, num_samples, pcm = read_wav_pcm(path, hop)
    c = pcm.astype(np.float64)
    mag = np.abs(c)
    return rate, n_ch, num_samples, np.where(mag >= 32768, 0.0, np.sign(c) * 2.0 * mag / 65535)


def encode_wav(in_path, out_path=None, use_huffman=True, device_id=0, handle=None, exact_spread=False, certify=None):
    """certify: a dict to fill with the sensitivity certificate of the encode (mrc_get_sensitivity: how many integer decisions
    were taken within a guard band of floating-point rounding); if any was, the file is encoded once more with the masker
    spreading evaluated operation by operation (MRC_OPT_EXACT_SPREAD) and certify["bytes_equal_exact_spread"] says whether
    the two encodes gave the same bytes.
    exact_spread: evaluate the masker spreading operation by operation like psychoac.py:68-78 (MRC_OPT_EXACT_SPREAD,
    ~30x slower kernel).  Both modes give the reference driver's bytes on every fixture and sweep; neither is
    bit-identical by construction (README.md, "Parity").
    The file's int16 codes go to the device as they are: the transient detector (mrc_transient_peaks_ex) and the whole
    encode loop (ONE call, mrc_encode_chained_stream_pcm16_pac) read them there; 2 bytes per sample on the host."""
    rate, n_ch, num_samples, pcm = read_wav_pcm(in_path)
    if n_ch != 2:
        raise ValueError("stereo input only (the reference's JointEncode indexes data[0], data[1])")
    h = handle if handle is not None else Handle(sample_rate=rate, device_id=device_id)
    was_exact = h.get_option(1)
    was_sens = h.get_option(5)
    if exact_spread:
        h.set_option(1, 1)
    if certify is not None:
        h.set_option(5, 1)
        h.sensitivity()
    try:
        L = h.cfg.n_mdct_lines
        codes = np.concatenate([np.zeros((2, L), np.int16), pcm], axis=1)      # the zero prior hop (pacfileThem.py:615-618)
        shapes = transient.block_shape_array(h, codes)
        if not len(shapes):
            raise ValueError("file too short: fewer than two hops")
        if shapes[-1, 2] != L:
            raise ValueError("the stream must end with a long block (the reference's Close() assumes it)")
        r = h.encode_chained_pac(codes[0][None], codes[1][None], [shapes], use_huffman=use_huffman, with_flush=True,
                                 num_samples=[num_samples])
        data = r["bytes"].tobytes()
        if certify is not None:
            certify.update(h.sensitivity())
            near = sum(certify[k] for k in ("quantiser_edges", "bitalloc_near_ties", "ms_switch_near_threshold", "peak_near_ties"))
            certify["decisions_near_an_edge"] = near
            if near and not exact_spread:
                h.set_option(5, 0)
                h.set_option(1, 1)
                again = h.encode_chained_pac(codes[0][None], codes[1][None], [shapes], use_huffman=use_huffman, with_flush=True,
                                             num_samples=[num_samples])
                certify["bytes_equal_exact_spread"] = again["bytes"].tobytes() == data
    finally:
        if handle is not None:
            h.set_option(1, was_exact)           # (a caller's handle gets back the settings it came with)
            h.set_option(5, was_sens)
        if handle is None:
            h.close()
    if out_path:
        with open(out_path, "wb") as f:
            f.write(data)
    return data


def wav_bytes(pcm, sample_rate):
    """pcmfile.py:141-153 header + interleaved little-endian int16 samples; pcm int16 [nCh][samples]."""
    n_ch, n = pcm.shape
    data = np.ascontiguousarray(pcm.T).astype("<i2").tobytes()
    head = pack('<4sL4s4sLHHLLHH4sL', b"RIFF", 36 + len(data), b"WAVE", b"fmt ", 16, 1, n_ch, sample_rate,
                sample_rate * n_ch * 2, n_ch * 2, 16, b"data", len(data))
    return head + data


def decode_pac_file(pac_path, wav_path, device_id=0):
    with open(pac_path, "rb") as fp:
        buf = fp.read()
    cfg, _, _, _ = pacfile.read_header(buf)
    h = Handle(sample_rate=cfg.sample_rate, n_mdct_lines=cfg.n_mdct_lines, n_scale_bits=cfg.n_scale_bits,
               n_mant_size_bits=cfg.n_mant_size_bits, device_id=device_id)
    try:
        pcm = pacfile.decode_pac_pcm16(h, buf)
    finally:
        h.close()
    data = wav_bytes(pcm, cfg.sample_rate)
    with open(wav_path, "wb") as fp:
        fp.write(data)
    return pcm


def main(argv=None):
    ap = argparse.ArgumentParser(description="Encode a stereo 16-bit WAV to .pac (or, with -d, decode a .pac to WAV) "
                                             "on an MI355X")
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("-d", "--decode", action="store_true")
    ap.add_argument("--no-huffman", action="store_true")
    ap.add_argument("--exact-spread", action="store_true",
                    help="masker spreading operation by operation as in psychoac.py:68-78 (slower kernel)")
    ap.add_argument("--certify", action="store_true",
                    help="report how many integer decisions of the encode lay within a guard band of floating-point rounding "
                         "(quantiser edges, bit-allocation ties, M/S threshold, peak test); if any did, encode again with "
                         "--exact-spread and say whether the bytes are the same")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if a.decode:
        pcm = decode_pac_file(a.src, a.dst, a.device)
        print("%s: %d channels x %d samples" % (a.dst, pcm.shape[0], pcm.shape[1]))
        return
    cert = {} if a.certify else None
    data = encode_wav(a.src, a.dst, not a.no_huffman, a.device, exact_spread=a.exact_spread, certify=cert)
    print("%s: %d bytes" % (a.dst, len(data)))
    if cert is not None:
        print("certificate: %d blocks examined; decisions within a guard band of rounding: %d (quantiser edges %d, "
              "bit-allocation ties %d, M/S threshold %d, peak test %d); chunks the slope-node evaluation sent back: %d"
              % (cert["blocks_examined"], cert["decisions_near_an_edge"], cert["quantiser_edges"], cert["bitalloc_near_ties"],
                 cert["ms_switch_near_threshold"], cert["peak_near_ties"], cert["node_chunks_sent_back"]))
        if "bytes_equal_exact_spread" in cert:
            print("  re-encoded with --exact-spread: bytes %s" % ("identical" if cert["bytes_equal_exact_spread"] else "DIFFER"))


if __name__ == "__main__":
    main()
