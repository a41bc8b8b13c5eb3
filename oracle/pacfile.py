"""
oracle.pacfile -- bit packing and `.pac` framing of encoded blocks (TEST ORACLE).

Restates bitpack.py:13-101 (PackedBits.WriteBits, MSB first), pacfileThem.py:586-619 (file header),
622-790 (WriteDataBlock: independent channels), 793-972 (JointWriteDataBlock), 973-984 (Close: one
extra NON-joint block of zeros) and the driver loop of pacfileThem.py:1159-1214 for a given sequence
of block shapes.  Pinned by tests/golden/ref_pac.npz -- the .pac files the reference's own driver
(pacfileThem.py run as a script through tests/golden/py2harness.py) wrote for synthetic WAV files, with and
without Huffman tables: encode_wav() reproduces them byte for byte (tests/test_reference_golden.py) -- and by the
bit-packer known-answer vector (bitpack.py:183-196: (3,5,11,3,1) in (4,3,5,3,1) bits -> 0x3A 0xB7).
Huffman table ids follow oracle.huffman_tables.TABLE_ORDER.
"""
from struct import pack

import numpy as np

from . import codec
from .huffman_tables import TABLES, TABLE_ORDER, RAW_TABLE_ID
from .psychoac import AssignMDCTLinesFromFreqLimits, ScaleFactorBands, py2div

BYTESIZE = 8


class PackedBits:
    """bitpack.py:13-101: fixed-size zeroed byte array filled MSB-first; WriteBits(info, n) appends the
    lowest n bits of info."""

    def __init__(self):
        self.iByte = self.iBit = 0

    def Size(self, nBytes):
        self.nBytes = int(nBytes)
        self.iByte = self.iBit = 0
        self.data = np.zeros(self.nBytes, dtype=np.uint8)

    def GetPackedData(self):
        return self.data.tobytes()

    def WriteBits(self, info, nBits):
        info = int(info)
        for i in range(int(nBits) - 1, -1, -1):         # same bytes as the reference's three-phase writer
            if (info >> i) & 1:
                self.data[self.iByte] += np.uint8(1 << (BYTESIZE - 1 - self.iBit))
            self.iBit += 1
            if self.iBit == BYTESIZE:
                self.iBit = 0
                self.iByte += 1


def file_header(cp, numSamples):
    """pacfileThem.py:586-613.  numSamples is padded only when it ALREADY is a multiple of nMDCTLines
    (inverted test, 595-597).  The band table in the header is always the long one."""
    if not numSamples % cp.nMDCTLines:
        numSamples += (cp.nMDCTLines - numSamples % cp.nMDCTLines)
    sfb = ScaleFactorBands(AssignMDCTLinesFromFreqLimits(cp.nMDCTLines, cp.sampleRate))
    out = b"PAC "
    out += pack('<LHLLHH', cp.sampleRate, cp.nChannels, numSamples, cp.nMDCTLines, cp.nScaleBits, cp.nMantSizeBits)
    out += pack('<L', sfb.nBands)
    out += pack('<' + str(sfb.nBands) + 'H', *(sfb.nLines.tolist()))
    return out


def _escape_code(table_id):
    table, esc = TABLES[TABLE_ORDER[table_id]]
    return table[esc][0]


def _mantissa_bits(bitAlloc, mantissa, huffTable, sfBands):
    """Bits of the band records' mantissa part as counted at pacfileThem.py:661-703 / 836-870."""
    bits = 0
    if huffTable == RAW_TABLE_ID:
        for iBand in range(sfBands.nBands):
            if bitAlloc[iBand]:
                bits += bitAlloc[iBand] * sfBands.nLines[iBand]
        return bits
    esc = _escape_code(huffTable)
    iMant = 0
    for iBand in range(sfBands.nBands):
        if bitAlloc[iBand]:
            for _ in range(int(sfBands.nLines[iBand])):
                code = str(mantissa[iMant])
                length = len(code.split("/")[0])
                bits += length
                if code[0:length] == esc:
                    bits += bitAlloc[iBand]
                iMant += 1
    return bits


def _write_band_records(pb, scaleFactor, bitAlloc, mantissa, huffTable, cp):
    """pacfileThem.py:727-781 / 909-963."""
    sfBands = cp.sfBands
    esc = None if huffTable == RAW_TABLE_ID else _escape_code(huffTable)
    iMant = 0
    for iBand in range(sfBands.nBands):
        ba = bitAlloc[iBand]
        pb.WriteBits(ba - 1 if ba else 0, cp.nMantSizeBits)
        pb.WriteBits(scaleFactor[iBand], cp.nScaleBits)
        if bitAlloc[iBand]:
            for _ in range(int(sfBands.nLines[iBand])):
                if esc is None:
                    pb.WriteBits(mantissa[iMant], bitAlloc[iBand])
                else:
                    code = str(mantissa[iMant])
                    head = code.split("/")[0]
                    for ch in head:
                        pb.WriteBits(int(ch), 1)
                    if head == esc:
                        pb.WriteBits(int(code.split("/")[1]), bitAlloc[iBand])
                iMant += 1


def _blksw_bit(length, cp):
    return 1 - py2div(int(length), int(cp.nMDCTLines))       # pacfileThem.py:720-721


def pack_block(scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable, cp):
    """Bytes WriteDataBlock appends for one block (pacfileThem.py:652-790): per channel `<L nBytes` + payload."""
    out = b""
    for iCh in range(cp.nChannels):
        nBits = cp.nScaleBits + 4
        nBits += cp.sfBands.nBands * (cp.nMantSizeBits + cp.nScaleBits)
        nBits += _mantissa_bits(bitAlloc[iCh], mantissa[iCh], huffTable[iCh], cp.sfBands)
        nBits += cp.blkswBitA + cp.blkswBitB
        nBytes = nBits // BYTESIZE if nBits % BYTESIZE == 0 else nBits // BYTESIZE + 1
        out += pack("<L", int(nBytes))
        pb = PackedBits()
        pb.Size(nBytes)
        pb.WriteBits(huffTable[iCh], 4)
        pb.WriteBits(_blksw_bit(cp.a, cp), cp.blkswBitA)
        pb.WriteBits(_blksw_bit(cp.b, cp), cp.blkswBitB)
        pb.WriteBits(overallScaleFactor[iCh], cp.nScaleBits)
        _write_band_records(pb, scaleFactor[iCh], bitAlloc[iCh], mantissa[iCh], huffTable[iCh], cp)
        out += pb.GetPackedData()
    return out


def pack_joint_block(scaleFactor, bitAlloc, mantissa, overallScaleFactor, ms_switch, huffTable, cp):
    """Bytes JointWriteDataBlock appends for one block (pacfileThem.py:825-970)."""
    out = b""
    for iCh in range(cp.nChannels):
        nBits = 0
        if iCh == 0:
            nBits += cp.sfBands.nBands + 4 * cp.nScaleBits
        nBits += 4
        nBits += cp.sfBands.nBands * (cp.nMantSizeBits + cp.nScaleBits)
        nBits += _mantissa_bits(bitAlloc[iCh], mantissa[iCh], huffTable[iCh], cp.sfBands)
        nBits += cp.blkswBitA + cp.blkswBitB
        nBytes = nBits // BYTESIZE if nBits % BYTESIZE == 0 else nBits // BYTESIZE + 1
        out += pack("<L", int(nBytes))
        pb = PackedBits()
        pb.Size(nBytes)
        pb.WriteBits(huffTable[iCh], 4)
        pb.WriteBits(_blksw_bit(cp.a, cp), cp.blkswBitA)
        pb.WriteBits(_blksw_bit(cp.b, cp), cp.blkswBitB)
        if iCh == 0:
            for s in overallScaleFactor:                        # L, R, M, S
                pb.WriteBits(s, cp.nScaleBits)
            for iBand in range(cp.sfBands.nBands):
                pb.WriteBits(ms_switch[iBand], 1)
        _write_band_records(pb, scaleFactor[iCh], bitAlloc[iCh], mantissa[iCh], huffTable[iCh], cp)
        out += pb.GetPackedData()
    return out


def encode_stereo_stream(stream, shapes, cp=None, huffman=True, num_samples=None):
    """The encode half of the reference CLI (pacfileThem.py:1105-1214 + Close 973-984) for a stereo
    stream that starts with the zero prior hop and a GIVEN sequence of block shapes [(offset, a, b)]
    (the transient detector is out of scope): header, one JointWriteDataBlock per shape with the bit
    reservoir chained through Huffman savings, then the flush block written by the NON-joint writer.
    Returns the .pac bytes."""
    cp = cp or codec.default_params(nChannels=2)
    if shapes[-1][2] != cp.nMDCTLines:
        raise ValueError("the stream must end with a long block: the reference's Close() pushes nMDCTLines zeros "
                         "through WriteDataBlock without updating b (pacfileThem.py:973-984) and fails otherwise")
    n_new = sum(b for (_, _, b) in shapes) if num_samples is None else num_samples   # the CLI writes the WAV's count
    out = file_header(cp, n_new)
    for (off, a, b) in shapes:
        cp.a, cp.b = a, b
        cp.sfBands = codec.bands_for_block(a, b, cp.nMDCTLines, cp.sampleRate)
        blk = [stream[0][off:off + a + b].copy(), stream[1][off:off + a + b].copy()]
        if huffman:
            r = codec.JointEncode(blk, cp)
        else:
            sf, ba, m, o, sw = codec.JointEncodeChannels(blk[0], blk[1], cp)
            r = (sf, ba, m, o, sw, [RAW_TABLE_ID, RAW_TABLE_ID])
        out += pack_joint_block(r[0], r[1], r[2], r[3], r[4], r[5], cp)
        last_hop = [x[off + a:off + a + b] for x in stream]
        cp.a = cp.b                                             # pacfileThem.py:1200,1209
    # Close (pacfileThem.py:973-984): zeros pushed through WriteDataBlock -> codec.Encode, channel by channel
    cp.b = cp.nMDCTLines
    cp.sfBands = codec.bands_for_block(cp.a, cp.b, cp.nMDCTLines, cp.sampleRate)
    blk = [np.concatenate([h, np.zeros(cp.nMDCTLines)]) for h in last_hop]
    r = codec.Encode(blk, cp) if huffman else codec.EncodeNoHuff(blk, cp)
    out += pack_block(r[0], r[1], r[2], r[3], r[4], cp)
    return out


def encode_wav(path, huffman=True):
    """The encode direction of `python pacfileThem.py in.wav` (pacfileThem.py:1084-1226) for a stereo 16-bit
    WAV: ingest (pcmfile), transient detection with one hop of look-ahead, joint blocks, Close().  Returns the
    .pac bytes.  As in the reference, the LAST hop of the file is analysed but never encoded."""
    from . import pcmfile, transient
    sampleRate, nChannels, numSamples, hops = pcmfile.read_wav(path)
    if nChannels != 2:
        raise ValueError("the reference CLI only works for stereo input (codecThem.py:264-265)")
    cp = codec.default_params(sampleRate=sampleRate, nChannels=2)
    stream = np.concatenate([np.zeros((2, cp.nMDCTLines)), hops], axis=1)
    shapes = transient.block_shapes(stream, cp)
    if not shapes:
        raise ValueError("file too short: the reference writes no block for fewer than two hops")
    cp.bitReservoir = 0
    return encode_stereo_stream(stream, shapes, cp, huffman, num_samples=numSamples)
