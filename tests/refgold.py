"""
Helpers for the fixtures tests/golden/ref_*.npz -- outputs of the reference's OWN hot-path functions, recorded by
tests/golden/make_golden_ref.py through tests/golden/py2harness.py (build container only).  The same replay code
drives the oracle (CPU tests) and the HIP drop-in module (GPU tests): `codec` is either `oracle.codec` or
`mrcaudiocodec_amd.codecThem`.
"""
import os
import types

import numpy as np

from oracle import psychoac as opsy            # band-table objects for codingParams (pinned by ref_psychoac.npz)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHORT_LIMITS = [300, 630, 1080, 1720, 2700, 4400, 7700, 15500, 24000]      # pacfileThem.py:643
SHAPES = [(1024, 1024), (1024, 128), (128, 128), (128, 1024)]


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def pcm_to_float(pcm):
    """pcmfile.py:91-100 (pinned by `pcmmap_*` in ref_encode.npz)."""
    p = np.asarray(pcm, dtype=np.float64)
    mag = np.abs(p)
    return np.where(mag >= 32768, 0.0, np.sign(p) * 2.0 * mag / 65535)


def bands(a, b, fs):
    half = (a + b) // 2
    if a + b == 2048:
        return opsy.ScaleFactorBands(opsy.AssignMDCTLinesFromFreqLimits(half, fs))
    return opsy.ScaleFactorBands(opsy.AssignMDCTLinesFromFreqLimits(half, fs, SHORT_LIMITS))


def params_from(e, tag):
    fs, nch, nscale, nmant, tbps = e[tag + "_params"]
    cp = types.SimpleNamespace()
    cp.sampleRate, cp.nChannels, cp.nMDCTLines = int(fs), int(nch), 1024
    cp.nScaleBits, cp.nMantSizeBits, cp.targetBitsPerSample = int(nscale), int(nmant), float(tbps)
    cp.nSamplesPerBlock, cp.bitReservoir, cp.nSamplesShort = 1024, 0, 128
    cp.a = cp.b = 1024
    cp.blkswBitA = cp.blkswBitB = 1
    return cp


def dense(m, ba, nlines):
    out = np.zeros(int(np.sum(nlines)), dtype=np.int64)
    lo = np.cumsum(nlines) - nlines
    i = 0
    for k in range(len(nlines)):
        if ba[k]:
            out[lo[k]:lo[k] + nlines[k]] = np.asarray(m[i:i + nlines[k]])
            i += nlines[k]
    assert i == len(m), "compact mantissa array has %d entries, bands with bits hold %d" % (len(m), i)
    return out


def blocks_of(e, tag):
    """-> list of (a, b, [channel arrays of a+b samples]) in stream order, framed as pacfileThem.py:628-631 does."""
    x = np.array([pcm_to_float(c) for c in e[tag + "_pcm"]])
    prior = np.zeros((x.shape[0], 1024))
    pos, out = 0, []
    for (a, b) in e[tag + "_shapes"]:
        a, b = int(a), int(b)
        new = x[:, pos:pos + b]
        pos += b
        out.append((a, b, [np.concatenate((prior[c][-a:], new[c])) for c in range(x.shape[0])]))
        prior = new
    return out


def check_chain(codec, e, tag, kind):
    """Replay one recorded chain through `codec` and compare every integer with the reference's.
    kind: 'single' (EncodeSingleChannel), 'nohuff' (EncodeNoHuff), 'indep' (Encode), 'jointch'
    (JointEncodeChannels), 'joint' (JointEncode).  The reservoir is carried by `codec` itself."""
    cp = params_from(e, tag)
    joint = kind in ("jointch", "joint")
    for i, (a, b, full) in enumerate(blocks_of(e, tag)):
        cp.a, cp.b, cp.sfBands = a, b, bands(a, b, cp.sampleRate)
        nl = np.asarray(cp.sfBands.nLines)
        k = "%s_%d" % (tag, i)
        where = "%s block %d shape (%d,%d)" % (tag, i, a, b)
        assert cp.bitReservoir == int(e[tag + "_res_in"][i]), "%s: reservoir in %d != %d" % (
            where, cp.bitReservoir, int(e[tag + "_res_in"][i]))
        data = [f.copy() for f in full]
        if kind == "single":
            sf, ba, mant, osf = codec.EncodeSingleChannel(data[0], cp)
            sf, ba, mant, osf, tbl = [sf], [ba], [mant], [osf], [15]
        elif kind == "nohuff":
            sf, ba, mant, osf, tbl = codec.EncodeNoHuff(data, cp)
        elif kind == "indep":
            sf, ba, mant, osf, tbl = codec.Encode(data, cp)
        elif kind == "jointch":
            sf, ba, mant, osf, ms = codec.JointEncodeChannels(data[0], data[1], cp)
            tbl = [15, 15]
        else:
            sf, ba, mant, osf, ms, tbl = codec.JointEncode(data, cp)
        assert cp.bitReservoir == int(e[tag + "_res_out"][i]), "%s: reservoir out %d != %d" % (
            where, cp.bitReservoir, int(e[tag + "_res_out"][i]))
        assert np.array_equal(np.asarray(osf, dtype=np.int64).ravel(), e[k + "_os"].ravel()), where + ": overall scale"
        assert np.array_equal(np.asarray(sf, dtype=np.int64), e[k + "_sf"]), where + ": scale factors"
        assert np.array_equal(np.asarray(ba, dtype=np.int64), e[k + "_ba"]), where + ": bit allocation"
        assert [int(t) for t in tbl] == [int(t) for t in e[k + "_table"]], where + ": Huffman table ids"
        if joint:
            assert np.array_equal(np.asarray(ms, dtype=np.int64), e[k + "_ms"]), where + ": M/S switch"
        for c in range(len(sf)):
            if int(tbl[c]) == 15:
                got = dense(np.asarray(mant[c]), np.asarray(ba[c]), nl)
                assert np.array_equal(got, e[k + "_mant%d" % c]), where + ": mantissas of stream %d" % c
            else:
                want = str(e[k + "_codes%d" % c]).split("\n")
                assert [str(v) for v in mant[c]] == want, where + ": Huffman code strings of stream %d" % c
    return cp
