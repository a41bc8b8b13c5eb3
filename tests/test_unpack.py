"""
CPU tests of the decode-side host code: the C++ `.pac` header / chunk parser (mrc_pac_read_header,
mrc_pac_scan_chunks, mrc_unpack_blocks; no GPU needed) against the oracle's restatement of the reference's reader
(oracle/decode.py), on streams the oracle's writer produced -- raw and Huffman-coded, with block switching.
"""
import numpy as np
import pytest

from mrcaudiocodec_amd import pacfile as ppac, synth
from oracle import decode as odec, pacfile as opac


def _stream(hops, switched):
    tone = synth.c1_sine(hops)
    if switched:
        x, shapes = synth.c4_transients(hops)
        return np.stack([x + 0.3 * tone, 0.7 * x + 0.3 * tone]), shapes
    g = synth.c2_noise(hops, seed=5, sigma=0.02)
    return np.stack([tone + g, 0.9 * tone - g]), [(i * 1024, 1024, 1024) for i in range(hops - 1)]


@pytest.mark.parametrize("huff", [False, True])
@pytest.mark.parametrize("switched", [False, True])
def test_unpack_matches_oracle_reader(huff, switched):
    stream, shapes = _stream(11 if switched else 6, switched)
    pac = opac.encode_stereo_stream(stream, shapes, huffman=huff)
    cfg, nch, num_samples, off = ppac.read_header(pac)
    cp, off_ref = odec.read_header(pac)
    assert (cfg.sample_rate, cfg.n_mdct_lines, cfg.n_scale_bits, cfg.n_mant_size_bits, nch, num_samples, off) == \
           (cp.sampleRate, cp.nMDCTLines, cp.nScaleBits, cp.nMantSizeBits, cp.nChannels, cp.numSamples, off_ref)
    chunks = ppac.scan_chunks(pac, off)
    ref_chunks = odec.split_chunks(pac, off_ref)
    assert len(chunks) == len(ref_chunks) == 2 * (len(shapes) + 1)
    n_joint = len(shapes)
    got = ppac.unpack_blocks(cfg, pac, chunks[:2 * n_joint], 2, True)
    for i, (_, a, b) in enumerate(shapes):
        want = odec.parse_joint_block(ref_chunks[2 * i], ref_chunks[2 * i + 1], cp)
        nb, half = cp.sfBands.nBands, (a + b) // 2
        assert (got["a"][i], got["b"][i]) == (a, b) == (cp.a, cp.b)
        assert list(got["huff_table"][i]) == want["huffTable"]
        assert list(got["overall_scale"][i]) == want["overallScale"]
        assert list(got["ms_switch"][i, :nb]) == want["ms_switch"] and not got["ms_switch"][i, nb:].any()
        for ch in range(2):
            assert list(got["scale_factor"][i, ch, :nb]) == want["scaleFactor"][ch]
            assert list(got["bit_alloc"][i, ch, :nb]) == want["bitAlloc"][ch]
            assert np.array_equal(got["mantissa"][i, ch, :half], want["mantissa"][ch][:half])
    if huff and not switched:
        assert (got["huff_table"] != 15).any()              # the Huffman branch is exercised
    flush = ppac.unpack_blocks(cfg, pac, chunks[2 * n_joint:], 2, False)
    for ch in range(2):
        want = odec.parse_block(ref_chunks[2 * n_joint + ch], cp)
        nb = cp.sfBands.nBands
        assert flush["overall_scale"][0, ch] == want["overallScale"] and flush["huff_table"][0, ch] == want["huffTable"]
        assert list(flush["bit_alloc"][0, ch, :nb]) == want["bitAlloc"]
        assert np.array_equal(flush["mantissa"][0, ch], want["mantissa"])


def test_unpack_rejects_damaged_input():
    stream, shapes = _stream(4, False)
    pac = opac.encode_stereo_stream(stream, shapes, huffman=True)
    cfg, nch, _, off = ppac.read_header(pac)
    with pytest.raises(ppac.MrcError):
        ppac.read_header(b"RIFF" + pac[4:])
    with pytest.raises(ppac.MrcError):
        ppac.scan_chunks(pac[:-3], off)                     # last chunk truncated
    chunks = ppac.scan_chunks(pac, off)
    bad = bytearray(pac)
    bad[chunks[0] + 4] = 0x4F                                # table id 4: not a table, not raw
    with pytest.raises(ppac.MrcError):
        ppac.unpack_blocks(cfg, bytes(bad), chunks[:2], 2, True)


def test_unpack_rejects_crafted_headers_and_widths():
    """The file header sizes allocations and picks field widths: values no encoder writes are refused (a crafted
    nMantSizeBits >= 5 would otherwise reach 17..256-bit mantissa reads and shifts)."""
    import struct
    stream, shapes = _stream(4, False)
    pac = opac.encode_stereo_stream(stream, shapes, huffman=False)

    def patched(offset, fmt, value):
        b = bytearray(pac)
        b[offset:offset + struct.calcsize(fmt)] = struct.pack(fmt, value)
        return bytes(b)
    # header: "PAC " <L rate <H nCh <L numSamples <L nMDCTLines <H nScaleBits <H nMantSizeBits <L nBands
    for off, fmt, value in ((4, "<L", 0), (8, "<H", 0), (8, "<H", 3), (14, "<L", 0), (14, "<L", 1000), (14, "<L", 1 << 20),
                            (18, "<H", 0), (18, "<H", 9), (20, "<H", 0), (20, "<H", 9)):
        with pytest.raises(ppac.MrcError):
            ppac.read_header(patched(off, fmt, value))
    # nMantSizeBits = 5 is a legal header (the training script used it) but lets a chunk claim up to 32 mantissa bits
    # per line; more than 16 is refused by the band-record reader
    cfg, nch, _, off = ppac.read_header(patched(20, "<H", 5))
    assert cfg.n_mant_size_bits == 5
    chunks = ppac.scan_chunks(pac, off)
    bad = bytearray(patched(20, "<H", 5))
    # first band record of chunk 0 starts after table(4) + blksw(2) + 4 overall scales(16) + 25 M/S bits = bit 47
    first = chunks[0] + 4
    bits = 47
    for i in range(5):                                         # bit allocation field := 0b11111 -> ba = 32
        byte, bit = first + (bits + i) // 8, 7 - (bits + i) % 8
        bad[byte] |= 1 << bit
    with pytest.raises(ppac.MrcError):
        ppac.unpack_blocks(cfg, bytes(bad), chunks[:2], 2, True)


@pytest.mark.parametrize("shape", [(1024, 1024), (128, 128), (1024, 128), (128, 1024)])
@pytest.mark.parametrize("huff", [False, True])
def test_pack_unpack_round_trip_random_blocks(shape, huff):
    # packer -> parser on random contents of every block shape: allocations 0 / 2..16, mantissas covering every
    # table entry, the escape values and beyond; joint and independent channels
    a, b = shape
    cfg = ppac.make_config()
    bands = ppac.band_table(cfg, a, b)
    nb, half, n = len(bands), (a + b) // 2, 24
    rng = np.random.default_rng(a + 7 * b + int(huff))
    line_band = np.repeat(np.arange(nb), bands)
    ba = rng.integers(0, 17, size=(n, 2, nb)).astype(np.int32)
    ba[ba == 1] = 0
    ba[:3] = 0                                               # blocks without any mantissa
    scale = np.array([1, 2, 3, 5, 9, 17, 33, 65, 70, 1 << 16])[rng.integers(0, 10, size=(n, 2, 1))]
    mant = (rng.integers(0, 1 << 16, size=(n, 2, half)) % scale).astype(np.int64)
    mant = np.minimum(mant, (1 << np.maximum(ba[:, :, line_band], 1)) - 1).astype(np.int32)
    mant[ba[:, :, line_band] == 0] = 0
    sf = rng.integers(0, 16, size=(n, 2, nb)).astype(np.int32)
    osc4 = rng.integers(0, 16, size=(n, 4)).astype(np.int32)
    sw = rng.integers(0, 2, size=(n, nb)).astype(np.int32)
    head = ppac.header(cfg, 2, n * b)

    data, offs, table, _ = ppac.pack_joint_blocks(cfg, a, b, osc4, sw, sf, ba, mant, huff)
    blob = head + data.tobytes()
    chunks = ppac.scan_chunks(blob, len(head))
    assert len(chunks) == 2 * n
    got = ppac.unpack_blocks(cfg, blob, chunks, 2, True)
    assert (got["a"] == a).all() and (got["b"] == b).all()
    assert np.array_equal(got["huff_table"], table)
    assert np.array_equal(got["overall_scale"], osc4) and np.array_equal(got["ms_switch"][:, :nb], sw)
    assert np.array_equal(got["scale_factor"][:, :, :nb], sf) and np.array_equal(got["bit_alloc"][:, :, :nb], ba)
    assert np.array_equal(got["mantissa"][:, :, :half], mant) and not got["mantissa"][:, :, half:].any()

    data, offs, table, _ = ppac.pack_blocks(cfg, a, b, osc4[:, :2], sf, ba, mant, huff)
    blob = head + data.tobytes()
    got = ppac.unpack_blocks(cfg, blob, ppac.scan_chunks(blob, len(head)), 2, False)
    assert np.array_equal(got["huff_table"], table) and np.array_equal(got["overall_scale"], osc4[:, :2])
    assert np.array_equal(got["scale_factor"][:, :, :nb], sf) and np.array_equal(got["bit_alloc"][:, :, :nb], ba)
    assert np.array_equal(got["mantissa"][:, :, :half], mant)
    if huff:
        assert (table != 15).any()
