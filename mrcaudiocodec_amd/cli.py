"""
Encode direction of the reference's command line (`python pacfileThem.py in.wav`, pacfileThem.py:1064-1231)
on the MI355X path:  python -m mrcaudiocodec_amd.cli in.wav out.pac [--no-huffman] [--device N]

WAV ingest (pcmfile.py:34-102: 16-bit PCM, int16 code c -> sign(c) 2|c|/65535), transient detection with
one hop of look-ahead (pacfileThem.py:1025-1056, 1182-1214), joint-stereo blocks with the bit reservoir
chained through the Huffman savings, Close()'s flush block, `.pac` framing -- kernels on the GPU, Huffman
and bit packing in C++ on the host.  Like the reference: stereo input only, and the last hop of the file is
analysed but never encoded.  Decoding is out of scope.
"""
import argparse
from struct import unpack

import numpy as np

from . import Handle, pacfile, transient


def read_wav(path, hop=1024):
    """-> (sample_rate, n_channels, num_samples, float64 [nCh][nHops*hop]), last hop zero padded."""
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
    c = np.frombuffer(raw, dtype="<i2").astype(np.float64)
    c = c[:(len(c) // n_ch) * n_ch].reshape(-1, n_ch).T
    n_hops = -(-c.shape[1] // hop)
    x = np.zeros((n_ch, n_hops * hop))
    mag = np.abs(c)
    x[:, :c.shape[1]] = np.where(mag >= 32768, 0.0, np.sign(c) * 2.0 * mag / 65535)    # -32768 -> 0.0 (pcmfile.py:91-100)
    return rate, n_ch, num_samples, x


def encode_wav(in_path, out_path=None, use_huffman=True, device_id=0, handle=None):
    rate, n_ch, num_samples, hops = read_wav(in_path)
    if n_ch != 2:
        raise ValueError("stereo input only (the reference's JointEncode indexes data[0], data[1])")
    h = handle if handle is not None else Handle(sample_rate=rate, device_id=device_id)
    try:
        stream = np.concatenate([np.zeros((2, h.cfg.n_mdct_lines)), hops], axis=1)
        shapes = transient.block_shapes(h, stream)
        if not shapes:
            raise ValueError("file too short: fewer than two hops")
        data = pacfile.encode_stereo_stream(h, stream, shapes, use_huffman, num_samples=num_samples)
    finally:
        if handle is None:
            h.close()
    if out_path:
        with open(out_path, "wb") as f:
            f.write(data)
    return data


def main(argv=None):
    ap = argparse.ArgumentParser(description="Encode a stereo 16-bit WAV to .pac on an MI355X")
    ap.add_argument("wav")
    ap.add_argument("pac")
    ap.add_argument("--no-huffman", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    data = encode_wav(a.wav, a.pac, not a.no_huffman, a.device)
    print("%s: %d bytes" % (a.pac, len(data)))


if __name__ == "__main__":
    main()
