"""
oracle.decode -- the decode side of the codec (TEST ORACLE for the "next" row f-4; not product code).

Restates, block by block like the reference:
    quantize.py:325-357  vDequantize        block floating point -> signed fraction
    mdct.py:98-122       IMDCT              N-point inverse FFT on the odd extension of the N/2 lines
    ms_stereo.py:33-49   ReconstructLR      per band L = M + S, R = M - S where the M/S switch is set
    codecThem.py:30-63   Decode             dequantise, undo the overall scale, IMDCT, transition window
    codecThem.py:65-134  JointDecode        same for a joint block: per band Mid/Left and Side/Right rescale levels
    bitpack.py:104-170   PackedBits.ReadBits
    pacfileThem.py:130-158, 161-319, 321-585   header, ReadDataBlock, JointReadDataBlock (chunk parsing, Huffman
                         prefix decoding, overlap-and-add)
    pcmfile.py:156-180   16-bit PCM codes of the decoded signed fractions

Pinned by golden vectors recorded from the reference's own py3-importable functions (vDequantize,
vDequantizeUniform, ReconstructLR: tests/golden/decode.npz) and by the TDAC known-answer vector of mdct.py:131-182;
IMDCT / Decode / JointDecode by tests/golden/ref_mdct.npz and ref_encode.npz (the reference's functions executed through
tests/golden/py2harness.py, bit-exact), and the chunk parser + overlap-and-add by ref_pac.npz: the WAV the
reference's own driver decoded equals this decoder's 16-bit codes from the second block on (its first written block
is a stale encode-direction buffer, see tests/golden/make_golden_pac.py).

Two deliberate differences from the reference's file layer, both documented in DESIGN.md:
  * Huffman codes are decoded from the code TABLES (the same prefix code the pickled trees hold; the tree pickles
    need the reference's HuffmanNode class to load) with the sorted-name table order the encoder side uses (F9).
  * The reference's CLI decode loop depends on state its encode loop left behind in the same process (which block
    is the non-joint one, a stale look-ahead block); here a stereo file is: joint blocks, then the two
    non-joint chunks Close() wrote (pacfileThem.py:973-984) -- which is what the encoder emits.
"""
from struct import unpack, calcsize

import numpy as np

from . import codec
from .huffman_tables import TABLES, TABLE_ORDER, RAW_TABLE_ID
from .psychoac import AssignMDCTLinesFromFreqLimits, ScaleFactorBands
from .quantize import vDequantizeUniform, vQuantizeUniform
from .window import TransitionWindow

SHORT_LIMITS = [300, 630, 1080, 1720, 2700, 4400, 7700, 15500, 24000]      # pacfileThem.py:214


def vDequantize(scale, mantissaVec, nScaleBits=3, nMantBits=5):
    """quantize.py:325-357."""
    cap = (1 << nScaleBits) - 1
    nBits = cap + nMantBits
    m = np.asarray(mantissaVec, dtype=np.float64)
    neg = m >= 2.0 ** (nMantBits - 1)
    mag = np.where(neg, m - 2.0 ** (nMantBits - 1), m)
    sign_code = np.where(neg, 2.0 ** (nBits - 1), 0.0)
    if scale == cap:
        quant = mag + sign_code
    else:
        shift = cap - int(scale)
        shifted = (mag.astype(np.uint64) << np.uint64(shift)).astype(np.float64)
        if shift > 0:
            shifted = shifted + np.where(mag > 0, 2.0 ** (shift - 1), 0.0)      # half a step, only for non-zero codes
        quant = shifted + sign_code
    return vDequantizeUniform(quant, nBits)


def IMDCT(data, a, b):
    """mdct.py:98-122."""
    N = a + b
    half = N // 2                                    # py2: N/2
    data = np.asarray(data, dtype=np.float64)
    X = np.zeros(N)
    X[0:half] = data
    X[half:] = -1 * data[::-1]
    n0 = (b + 1) / 2.0
    k = np.arange(N)
    pre = np.exp(np.multiply(k, 1j * 2 * np.pi * n0 / N))
    y = np.fft.ifft(np.multiply(pre, X), N)
    post = np.exp(np.multiply(np.add(k, n0), (1j * 2 * np.pi / (2.0 * N))))
    return N * np.real(np.multiply(y, post))


def ReconstructLR(mdct1, mdct2, sfBands, ms_switch):
    """ms_stereo.py:33-49."""
    left, right = np.array(mdct1, dtype=np.float64), np.array(mdct2, dtype=np.float64)
    for i in range(sfBands.nBands):
        lo, hi = sfBands.lowerLine[i], sfBands.upperLine[i] + 1
        if ms_switch[i] == 1:
            left[lo:hi] = mdct1[lo:hi] + mdct2[lo:hi]
            right[lo:hi] = mdct1[lo:hi] - mdct2[lo:hi]
    return left, right


def _dequantise_lines(scaleFactor, bitAlloc, mantissa, cp):
    halfN = (cp.a + cp.b) // 2
    line = np.zeros(halfN, dtype=np.float64)
    i = 0
    for iBand in range(cp.sfBands.nBands):
        n = int(cp.sfBands.nLines[iBand])
        if bitAlloc[iBand]:
            line[i:i + n] = vDequantize(scaleFactor[iBand], mantissa[i:i + n], cp.nScaleBits, bitAlloc[iBand])
        i += n
    return line


def Decode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, cp):
    """codecThem.py:30-63: one channel's windowed block of a + b samples (before overlap-and-add)."""
    line = _dequantise_lines(scaleFactor, bitAlloc, mantissa, cp)
    line /= 1. * (1 << overallScaleFactor)
    return TransitionWindow(IMDCT(line, cp.a, cp.b), cp.a, cp.b)


def JointDecode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, cp, ms_switch):
    """codecThem.py:65-134; overallScaleFactor = [L, R, M, S]."""
    lvl = [1. * (1 << int(s)) for s in overallScaleFactor]
    l1 = _dequantise_lines(scaleFactor[0], bitAlloc[0], mantissa[0], cp)
    l2 = _dequantise_lines(scaleFactor[1], bitAlloc[1], mantissa[1], cp)
    for iBand in range(cp.sfBands.nBands):
        lo, hi = cp.sfBands.lowerLine[iBand], cp.sfBands.upperLine[iBand] + 1
        ms = ms_switch[iBand] == 1
        if bitAlloc[0][iBand]:
            l1[lo:hi] /= lvl[2] if ms else lvl[0]
        if bitAlloc[1][iBand]:
            l2[lo:hi] /= lvl[3] if ms else lvl[1]
    left, right = ReconstructLR(l1, l2, cp.sfBands, ms_switch)
    return [TransitionWindow(IMDCT(left, cp.a, cp.b), cp.a, cp.b),
            TransitionWindow(IMDCT(right, cp.a, cp.b), cp.a, cp.b)]


# ---------------------------------------------------------------------------------------------- file layer
class BitReader:
    """bitpack.py:104-170 (ReadBits, MSB first) over a bytes object."""

    def __init__(self, data):
        self.data = np.frombuffer(data, dtype=np.uint8)
        self.pos = 0

    def ReadBits(self, nBits):
        v = 0
        for _ in range(int(nBits)):
            byte = int(self.data[self.pos >> 3])
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v


def read_header(buf):
    """pacfileThem.py:130-158 -> (CodingParams, offset of the first chunk)."""
    if buf[:4] != b"PAC ":
        raise ValueError("not a PAC file")
    fmt = '<LHLLHH'
    sampleRate, nChannels, numSamples, nMDCTLines, nScaleBits, nMantSizeBits = unpack(fmt, buf[4:4 + calcsize(fmt)])
    off = 4 + calcsize(fmt)
    nBands = unpack('<L', buf[off:off + 4])[0]
    off += 4
    nLines = unpack('<' + str(nBands) + 'H', buf[off:off + 2 * nBands])
    off += 2 * nBands
    cp = codec.CodingParams()
    cp.sampleRate, cp.nChannels, cp.numSamples = sampleRate, nChannels, numSamples
    cp.nMDCTLines = cp.nSamplesPerBlock = nMDCTLines
    cp.nScaleBits, cp.nMantSizeBits = nScaleBits, nMantSizeBits
    cp.sfBands = ScaleFactorBands(nLines)
    cp.nSamplesShort = 128
    cp.a = cp.b = nMDCTLines
    cp.blkswBitA = cp.blkswBitB = 1
    return cp, off


def split_chunks(buf, off):
    """The `<L nBytes` + payload chunks that follow the header."""
    chunks = []
    while off + 4 <= len(buf):
        n = unpack('<L', buf[off:off + 4])[0]
        if off + 4 + n > len(buf):
            raise ValueError("truncated PAC chunk")
        chunks.append(buf[off + 4:off + 4 + n])
        off += 4 + n
    return chunks


def _decode_table(table_id):
    name = TABLE_ORDER[table_id]
    table, escape_value = TABLES[name]
    rev = {code: value for value, (code, _len) in table.items()}
    return rev, table[escape_value][0]


def _read_prefix(pb, cp):
    """huffTable, block-switch bits -> cp.a, cp.b, cp.sfBands (pacfileThem.py:196-216)."""
    huffTable = pb.ReadBits(4)
    swA, swB = pb.ReadBits(cp.blkswBitA), pb.ReadBits(cp.blkswBitB)
    cp.a = (1 - swA) * cp.nMDCTLines + swA * 128
    cp.b = (1 - swB) * cp.nMDCTLines + swB * 128
    half = (cp.a + cp.b) // 2
    if cp.a + cp.b == 2 * cp.nMDCTLines:
        cp.sfBands = ScaleFactorBands(AssignMDCTLinesFromFreqLimits(half, cp.sampleRate))
    else:
        cp.sfBands = ScaleFactorBands(AssignMDCTLinesFromFreqLimits(half, cp.sampleRate, SHORT_LIMITS))
    return huffTable


def _read_band_records(pb, huffTable, cp):
    """pacfileThem.py:219-302: per band {ba-1 | 0, scale factor, mantissas or Huffman codes}.  Mantissas land at the
    band's own lines (dense layout)."""
    scaleFactor, bitAlloc = [], []
    mantissa = np.zeros(cp.nMDCTLines, np.int32)
    rev = esc = None
    if huffTable != RAW_TABLE_ID:
        rev, esc = _decode_table(huffTable)
    for iBand in range(cp.sfBands.nBands):
        ba = pb.ReadBits(cp.nMantSizeBits)
        if ba:
            ba += 1
        bitAlloc.append(ba)
        scaleFactor.append(pb.ReadBits(cp.nScaleBits))
        if not ba:
            continue
        lo = int(cp.sfBands.lowerLine[iBand])
        for j in range(int(cp.sfBands.nLines[iBand])):
            if rev is None:
                mantissa[lo + j] = pb.ReadBits(ba)
            else:
                code = ""
                while code not in rev:                  # walk the prefix code (the reference walks its tree)
                    code += "1" if pb.ReadBits(1) else "0"
                    if len(code) > 32:
                        raise ValueError("bad Huffman code in PAC chunk")
                mantissa[lo + j] = pb.ReadBits(ba) if code == esc else rev[code]
    return scaleFactor, bitAlloc, mantissa


def parse_block(chunk, cp):
    """One non-joint channel chunk (ReadDataBlock's inner part) -> dict; sets cp.a, cp.b, cp.sfBands."""
    pb = BitReader(chunk)
    huffTable = _read_prefix(pb, cp)
    overall = pb.ReadBits(cp.nScaleBits)
    sf, ba, mant = _read_band_records(pb, huffTable, cp)
    return dict(huffTable=huffTable, overallScale=overall, scaleFactor=sf, bitAlloc=ba, mantissa=mant)


def parse_joint_block(chunk0, chunk1, cp):
    """The two chunks of a joint block (JointReadDataBlock)."""
    pb = BitReader(chunk0)
    ht0 = _read_prefix(pb, cp)
    overall = [pb.ReadBits(cp.nScaleBits) for _ in range(4)]
    ms_switch = [pb.ReadBits(1) for _ in range(cp.sfBands.nBands)]
    sf0, ba0, m0 = _read_band_records(pb, ht0, cp)
    pb = BitReader(chunk1)
    ht1 = _read_prefix(pb, cp)
    sf1, ba1, m1 = _read_band_records(pb, ht1, cp)
    return dict(huffTable=[ht0, ht1], overallScale=overall, ms_switch=ms_switch, scaleFactor=[sf0, sf1],
                bitAlloc=[ba0, ba1], mantissa=[m0, m1])


def decode_pac(buf):
    """Decode a whole `.pac` byte string -> (cp, float64 [nCh][samples]): the concatenation of what successive
    (Joint)ReadDataBlock calls return, final overlap-and-add tail included.  The first block's output is the
    half-block delay of the MDCT (zeros overlap) and is kept here; pcm16() drops it like the reference's loop."""
    cp, off = read_header(buf)
    chunks = split_chunks(buf, off)
    nCh = cp.nChannels
    if len(chunks) % nCh:
        raise ValueError("chunk count is not a multiple of the channel count")
    nBlocks = len(chunks) // nCh
    overlap = [np.zeros(cp.nMDCTLines) for _ in range(nCh)]
    out = [[] for _ in range(nCh)]
    for blk in range(nBlocks):
        joint = nCh == 2 and blk < nBlocks - 1          # Close() flushes through the non-joint writer
        if joint:
            p = parse_joint_block(chunks[2 * blk], chunks[2 * blk + 1], cp)
            dec = JointDecode(p["scaleFactor"], p["bitAlloc"], p["mantissa"], p["overallScale"], cp, p["ms_switch"])
        else:
            dec = []
            for ch in range(nCh):
                p = parse_block(chunks[nCh * blk + ch], cp)
                dec.append(Decode(p["scaleFactor"], p["bitAlloc"], p["mantissa"], p["overallScale"], cp))
        for ch in range(nCh):
            out[ch].append(np.add(overlap[ch], dec[ch][:cp.a]))      # pacfileThem.py:312-315
            overlap[ch] = dec[ch][cp.a:]
    for ch in range(nCh):
        out[ch].append(overlap[ch])                     # the last call returns the pending half
    return cp, np.stack([np.concatenate(o) for o in out])


def pcm16(x):
    """pcmfile.py:163-172: signed fractions -> int16 codes (sign-magnitude quantiser, then 2's complement)."""
    x = np.asarray(x, dtype=np.float64)
    neg = np.signbit(x)
    code = vQuantizeUniform(np.abs(x), 16).astype(np.int16)
    return np.where(neg, -code, code).astype(np.int16)
