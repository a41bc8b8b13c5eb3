"""
Host-side back end (Huffman table choice + .pac bit packing, C++ in libmrc_hip.so, no GPU needed):
byte-for-byte against the oracle's restatement of pacfileThem.WriteDataBlock / JointWriteDataBlock /
WriteFileHeader and codecThem.calculateHuffmanGain, on blocks encoded by the oracle.  CPU only.
"""
import numpy as np
import pytest

from mrcaudiocodec_amd import pacfile as ppac, synth
from oracle import codec as ocodec, fast, pacfile as opac


def _cp(nch, a, b):
    cp = ocodec.default_params(nChannels=nch)
    cp.a, cp.b = a, b
    cp.sfBands = ocodec.bands_for_block(a, b, 1024, 48000)
    return cp


def test_band_table_and_header():
    cfg = ppac.make_config()
    for (a, b) in [(1024, 1024), (128, 128), (1024, 128), (128, 1024)]:
        assert list(ppac.band_table(cfg, a, b)) == list(fast.bands_for(a, b).nLines)
    for n in (5000, 4096, 0):
        assert ppac.header(cfg, 2, n) == opac.file_header(_cp(2, 1024, 1024), n)


@pytest.mark.parametrize("huff", [False, True])
@pytest.mark.parametrize("ab", [(1024, 1024), (128, 128), (1024, 128)])
def test_pack_independent_channels(ab, huff):
    a, b = ab
    x = synth.c1_sine(12) if huff else synth.c2_noise(12)      # a sine leaves mostly tiny mantissas: a Huffman table wins
    blocks = np.stack([x[s:s + a + b] for s in (0, 700, 2048, 5000)])
    r = fast.encode_mono_batch(blocks, a, b)
    cfg = ppac.make_config()
    data, offs, table, saved = ppac.pack_blocks(cfg, a, b, r["overall_scale"][:, None], r["scale_factor"][:, None, :],
                                                r["bit_alloc"][:, None, :], r["mantissa"][:, None, :], use_huffman=huff)
    cp = _cp(1, a, b)
    for i in range(blocks.shape[0]):
        m = fast.compact_mantissa(r["mantissa"][i], r["bit_alloc"][i], cp.sfBands)
        if huff:
            t, codes, sv = ocodec.calculateHuffmanGain(m, r["bit_alloc"][i], cp)
        else:
            t, codes, sv = 15, m, 0
        want = opac.pack_block([r["scale_factor"][i]], [r["bit_alloc"][i]], [codes], [int(r["overall_scale"][i])], [t], cp)
        assert data[offs[i]:offs[i + 1]].tobytes() == want, (i, t)
        assert table[i, 0] == t and saved[i, 0] == sv
    if huff:
        assert (table != 15).any()                             # the Huffman branch is exercised


@pytest.mark.parametrize("huff", [False, True])
def test_pack_joint(huff):
    s = synth.c3_stereo(6)
    if huff:                                                   # near-identical sines: tonal table wins, M/S on
        t = synth.c1_sine(6)
        s = np.stack([t, 0.9 * t + 1e-4 * s[1]])
    bl, br = np.array(fast.blocks_from_stream(s[0], 1024)), np.array(fast.blocks_from_stream(s[1], 1024))
    r = fast.encode_joint_batch(bl, br, 1024, 1024)
    cfg = ppac.make_config()
    data, offs, table, saved = ppac.pack_joint_blocks(cfg, 1024, 1024, r["overall_scale"], r["ms_switch"], r["scale_factor"],
                                                      r["bit_alloc"], r["mantissa"], use_huffman=huff)
    cp = _cp(2, 1024, 1024)
    for i in range(bl.shape[0]):
        ms, ts = [], []
        for c in range(2):
            m = fast.compact_mantissa(r["mantissa"][i, c], r["bit_alloc"][i, c], cp.sfBands)
            t, codes, sv = ocodec.calculateHuffmanGain(m, r["bit_alloc"][i, c], cp) if huff else (15, m, 0)
            ms.append(codes); ts.append(t)
            assert table[i, c] == t and saved[i, c] == sv
        want = opac.pack_joint_block(list(r["scale_factor"][i]), list(r["bit_alloc"][i]), ms,
                                     [int(v) for v in r["overall_scale"][i]], list(r["ms_switch"][i]), ts, cp)
        assert data[offs[i]:offs[i + 1]].tobytes() == want, i


def test_pack_threads_give_identical_bytes():
    s = synth.c3_stereo(40)
    bl, br = np.array(fast.blocks_from_stream(s[0], 1024)), np.array(fast.blocks_from_stream(s[1], 1024))
    r = fast.encode_joint_batch(bl, br, 1024, 1024)
    cfg = ppac.make_config()
    args = (cfg, 1024, 1024, r["overall_scale"], r["ms_switch"], r["scale_factor"], r["bit_alloc"], r["mantissa"])
    before = ppac.get_threads()
    assert 1 <= before <= 16                                   # default: the CPUs of the process, at most 16
    try:
        ppac.set_threads(1)
        one = ppac.pack_joint_blocks(*args)
        ppac.set_threads(5)
        many = ppac.pack_joint_blocks(*args)
    finally:
        ppac.set_threads(before)
    assert one[0].tobytes() == many[0].tobytes() and np.array_equal(one[1], many[1])
    assert np.array_equal(one[2], many[2]) and np.array_equal(one[3], many[3])


def test_pack_escape_and_buffer_checks():
    # hand-made block: values outside every table and the escape value itself (priced without its raw bits,
    # written with them: codecThem.py:169-172 vs 194-200)
    cfg = ppac.make_config()
    cp = _cp(1, 1024, 1024)
    nl = cp.sfBands.nLines
    ba = np.zeros(25, dtype=np.int32); ba[0] = 6; ba[1] = 5
    m = np.zeros(1024, dtype=np.int32)
    m[:4] = [16, 40, 0, 1]; m[4:9] = [0, 0, 7, 0, 2]
    sf = np.arange(25, dtype=np.int32) % 16
    data, offs, table, saved = ppac.pack_blocks(cfg, 1024, 1024, [[3]], sf[None, None, :], ba[None, None, :], m[None, None, :])
    comp = fast.compact_mantissa(m, ba, cp.sfBands)
    t, codes, sv = ocodec.calculateHuffmanGain(comp, ba, cp)
    assert table[0, 0] == t and saved[0, 0] == sv and t != 15
    assert data.tobytes() == opac.pack_block([sf], [ba], [codes], [3], [t], cp)
    from mrcaudiocodec_amd._lib import lib, MrcError
    with pytest.raises(MrcError):
        ppac.band_table(cfg, 0, 5)


def test_pack_with_given_tables_equals_priced():
    """mrc_pack_*_with_tables: the table ids chosen elsewhere (on the device by huffman_gain_kernel) give the same
    bytes as the host's own pricing; any valid id packs (forced tables), an invalid one is refused."""
    t = synth.c1_sine(8)
    n = synth.c3_stereo(8)
    s = np.stack([t + 0.01 * n[0], 0.9 * t + 1e-4 * n[1]])
    bl, br = np.array(fast.blocks_from_stream(s[0], 1024)), np.array(fast.blocks_from_stream(s[1], 1024))
    r = fast.encode_joint_batch(bl, br, 1024, 1024)
    cfg = ppac.make_config()
    args = (cfg, 1024, 1024, r["overall_scale"], r["ms_switch"], r["scale_factor"], r["bit_alloc"], r["mantissa"])
    data, offs, table, _ = ppac.pack_joint_blocks(*args, use_huffman=True)
    assert (table != 15).any()
    given = ppac.pack_joint_blocks(*args, huff_table=table)
    assert given[0].tobytes() == data.tobytes() and np.array_equal(given[1], offs)
    raw = ppac.pack_joint_blocks(*args, use_huffman=False)
    assert ppac.pack_joint_blocks(*args, huff_table=np.full_like(table, 15))[0].tobytes() == raw[0].tobytes()
    # a forced table (not the cheapest): still what the oracle's writer emits for that table id
    cp = _cp(2, 1024, 1024)
    forced = np.full_like(table, 2)
    got = ppac.pack_joint_blocks(*args, huff_table=forced)
    from oracle.huffman_tables import TABLES, TABLE_ORDER
    tab, esc = TABLES[TABLE_ORDER[2]]
    for i in range(bl.shape[0]):
        ms = []
        for c in range(2):
            m = fast.compact_mantissa(r["mantissa"][i, c], r["bit_alloc"][i, c], cp.sfBands)
            ms.append([tab[int(v)][0] if (int(v) in tab and int(v) != esc) else tab[esc][0] + "/" + str(int(v)) for v in m])
        want = opac.pack_joint_block(list(r["scale_factor"][i]), list(r["bit_alloc"][i]), ms,
                                     [int(v) for v in r["overall_scale"][i]], list(r["ms_switch"][i]), [2, 2], cp)
        assert got[0][got[1][i]:got[1][i + 1]].tobytes() == want, i
    mono = fast.encode_mono_batch(bl, 1024, 1024)
    margs = (cfg, 1024, 1024, mono["overall_scale"][:, None], mono["scale_factor"][:, None, :], mono["bit_alloc"][:, None, :],
             mono["mantissa"][:, None, :])
    d1, o1, t1, _ = ppac.pack_blocks(*margs, use_huffman=True)
    assert ppac.pack_blocks(*margs, huff_table=t1)[0].tobytes() == d1.tobytes()
    from mrcaudiocodec_amd._lib import MrcError
    with pytest.raises(MrcError):
        ppac.pack_joint_blocks(*args, huff_table=np.full_like(table, 7))


def test_pack_accepts_the_16_bit_mantissa_plane():
    """The PCM16 / mantissa16 encode paths deliver uint16 codes: the packer takes them as they are (also through an
    int16 view, which is how torch holds them) and writes the same bytes as from the int32 plane."""
    s = synth.c3_stereo(10)
    bl, br = np.array(fast.blocks_from_stream(s[0], 1024)), np.array(fast.blocks_from_stream(s[1], 1024))
    r = fast.encode_joint_batch(bl, br, 1024, 1024)
    # make some codes use all 16 bits (sign bit of a 16-bit allocation set: not representable as int16)
    r["bit_alloc"][0, 0, 3] = 16
    lo = int(np.sum(fast.bands_for(1024, 1024).nLines[:3]))
    r["mantissa"][0, 0, lo:lo + 4] = [0x8000, 0xFFFF, 0x8001, 0x7FFF]
    assert r["mantissa"].max() > 32767
    cfg = ppac.make_config()
    args = (cfg, 1024, 1024, r["overall_scale"], r["ms_switch"], r["scale_factor"], r["bit_alloc"])
    want = ppac.pack_joint_blocks(*args, r["mantissa"].astype(np.int32), use_huffman=True)
    m16 = r["mantissa"].astype(np.uint16)
    for plane in (m16, m16.view(np.int16)):
        got = ppac.pack_joint_blocks(*args, plane, use_huffman=True)
        assert got[0].tobytes() == want[0].tobytes() and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    mono = fast.encode_mono_batch(bl, 1024, 1024)
    margs = (cfg, 1024, 1024, mono["overall_scale"][:, None], mono["scale_factor"][:, None, :], mono["bit_alloc"][:, None, :])
    assert ppac.pack_blocks(*margs, mono["mantissa"][:, None, :].astype(np.uint16))[0].tobytes() == \
        ppac.pack_blocks(*margs, mono["mantissa"][:, None, :].astype(np.int32))[0].tobytes()
