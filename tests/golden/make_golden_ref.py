#!/usr/bin/env python3
"""
Golden vectors from the reference's OWN hot-path code (psychoac.py, window.py, mdct.py, codecThem.py ...),
executed in the build container through tests/golden/py2harness.py (Python-2 `/`, float sizes / indices and
`dict.has_key` given their Python 2 / NumPy<1.12 meaning; nothing else touched).  Outputs are data only:

    python tests/golden/make_golden_ref.py [/root/reference]     ->  tests/golden/ref_*.npz

ref_psychoac.npz  SPL / Intensity / Thresh / Bark on the MDCT grids, Masker.vIntensityAtBark (levels -30..96 dB,
                  |dz| == 0.5 exactly), AssignMDCTLinesFromFreqLimits + ScaleFactorBands at 48 / 44.1 kHz
                  (psychoac.py:8-131), CalcSMRs' post-threshold arithmetic
ref_window.npz    KBDWindow / TransitionWindow tables and windowed noise (window.py:49-121)
ref_mdct.npz      MDCT / IMDCT on noise blocks of every block shape (mdct.py:53-122)
ref_smr.npz       getMaskedThreshold / CalcSMRs per block shape and sample rate (psychoac.py:134-219)
ref_encode.npz    EncodeSingleChannel / JointEncodeChannels / Encode / EncodeNoHuff / JointEncode chains with
                  block switching, reservoir carried (codecThem.py:136-354, 359-574), and Decode / JointDecode
                  of the same blocks (codecThem.py:30-134)
Inputs are synthetic 16-bit PCM (seeded), stored in the files.  The Huffman tables the reference's
calculateHuffmanGain loads from ./training_data are written HERE from this repo's table data (the reference's
pickles are never unpickled); the directory order that os.walk/glob produced is recorded in the fixture.
"""
import copy
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

import py2harness as H                      # noqa: E402
from oracle import huffman_tables as HT      # noqa: E402  (table DATA only: value -> code string)

R = H.load_reference(REF)
rw, rmdct, rq, rp, rc = R["window"], R["mdct"], R["quantize"], R["psychoac"], R["codecThem"]
rng = np.random.default_rng(20261005)
SHORT_LIMITS = [300, 630, 1080, 1720, 2700, 4400, 7700, 15500, 24000]      # pacfileThem.py:643
SHAPES = [(1024, 1024), (1024, 128), (128, 128), (128, 1024)]


def pcm_to_float(pcm):
    """pcmfile.py:91-100 through the reference's own vDequantizeUniform."""
    codes = np.asarray([int(v) for v in pcm])          # pcmfile.py:93: array of the unpacked short ints
    signs = np.signbit(codes)
    codes[signs] *= -1
    temp = rq.vDequantizeUniform(codes, 16)
    temp[signs] *= -1.
    return temp


def gauss_pcm(n, sigma):
    return np.clip(np.rint(rng.normal(0, sigma * 32767, n)), -32767, 32767).astype(np.int16)


def bands(a, b, fs):
    half = (a + b) // 2
    if a + b == 2048:
        return rp.ScaleFactorBands(rp.AssignMDCTLinesFromFreqLimits(half, fs))
    return rp.ScaleFactorBands(rp.AssignMDCTLinesFromFreqLimits(half, fs, SHORT_LIMITS))


# ------------------------------------------------------------------------------------------------ psychoac
p = {}
grids = []
for fs in (48000, 44100):
    for half in (1024, 576, 128):
        f = (np.arange(half) + 0.5) * ((float(fs) / half) / 2.)
        key = "%d_%d" % (half, fs)
        grids.append((half, fs))
        p["freq_" + key] = f
        p["thresh_" + key] = rp.Thresh(f)
        p["bark_" + key] = rp.Bark(f)
        p["quiet_" + key] = rp.Intensity(rp.Thresh(f))
        p["spl_quiet_" + key] = rp.SPL(rp.Intensity(rp.Thresh(f)))
p["grids"] = np.array(grids)
edge_i = np.concatenate([[0.0, 1e-300, 1e-30, 2.5118864315095823e-13, 1e-12, 1e-9, 0.5, 1.0, 2.0, 1e6],
                         10.0 ** rng.uniform(-14, 2, 64)])
with np.errstate(divide="ignore"):
    p["spl_in"], p["spl_out"] = edge_i, rp.SPL(edge_i)
edge_s = np.concatenate([[-30.0, -29.999, 0.0, 40.0, 95.999, 96.0, 120.0], rng.uniform(-40, 110, 64)])
p["int_in"], p["int_out"] = edge_s, rp.Intensity(edge_s)
edge_f = np.concatenate([[11.71875, 23.4375, 100.0, 999.9, 1000.0, 3300.0, 7500.0, 20000.0, 23988.28125],
                         rng.uniform(10, 24000, 64)])
p["f_in"], p["bark_out"], p["thresh_out"] = edge_f, rp.Bark(edge_f), rp.Thresh(edge_f)
# maskers: level sweep x Bark position, evaluated on the long 48 kHz grid and on a grid with |dz| == 0.5 exactly
zgrid = rp.Bark(p["freq_1024_48000"])
mk_f, mk_spl, mk_out, mk_edge_z, mk_edge_out = [], [], [], [], []
for spl in (-30.0, -12.5, 0.0, 25.0, 39.999, 40.0, 40.001, 55.5, 70.0, 96.0, 110.0):
    for f in (46.0, 230.0, 1000.0, 2875.4, 9000.0, 15250.0, 21200.0):
        m = rp.Masker(f, spl)
        mk_f.append(f); mk_spl.append(spl)
        mk_out.append(m.vIntensityAtBark(zgrid))
        ze = m.z + np.array([-3.0, -0.5000000001, -0.5, -0.25, 0.0, 0.25, 0.5, 0.5000000001, 3.0, 11.0])
        mk_edge_z.append(ze)
        mk_edge_out.append(m.vIntensityAtBark(ze))
p["mk_f"], p["mk_spl"], p["mk_zgrid"] = np.array(mk_f), np.array(mk_spl), zgrid
p["mk_out"], p["mk_edge_z"], p["mk_edge_out"] = np.array(mk_out), np.array(mk_edge_z), np.array(mk_edge_out)
# band tables
for fs in (48000, 44100, 32000):
    for half, lim in ((1024, None), (576, SHORT_LIMITS), (128, SHORT_LIMITS), (1024, SHORT_LIMITS), (512, None)):
        nl = rp.AssignMDCTLinesFromFreqLimits(half, fs) if lim is None else rp.AssignMDCTLinesFromFreqLimits(half, fs, lim)
        sfb = rp.ScaleFactorBands(nl)
        key = "%d_%d_%s" % (half, fs, "cb" if lim is None else "short")
        p["bt_assign_" + key] = np.asarray(nl, dtype=np.float64)
        p["bt_lower_" + key], p["bt_upper_" + key], p["bt_nlines_" + key] = sfb.lowerLine, sfb.upperLine, sfb.nLines
# CalcSMRs' own arithmetic after the threshold (psychoac.py:212-217): the module's getMaskedThreshold is
# replaced, for this call only, by a function returning a recorded threshold
sfb = bands(1024, 1024, 48000)
lines = rng.normal(0, 1, 1024) * 10.0 ** rng.uniform(-6, 0, 1024)
lines[5] = 0.0
thr = rng.uniform(-20, 70, 1024)
orig = rp.getMaskedThreshold
rp.getMaskedThreshold = lambda *a, **k: thr
try:
    with np.errstate(divide="ignore"):
        p["smrpost_out"] = np.array([rp.CalcSMRs(np.zeros(2048), lines, sc, 48000, sfb) for sc in (0, 3, 15)])
finally:
    rp.getMaskedThreshold = orig
p["smrpost_lines"], p["smrpost_thr"], p["smrpost_scales"] = lines, thr, np.array([0, 3, 15])
np.savez_compressed(os.path.join(HERE, "ref_psychoac.npz"), **p)

# ------------------------------------------------------------------------------------------------ window
w = {}
for N in (2048, 256, 16):
    w["kbd_%d" % N] = rw.KBDWindow(np.ones(N))
for (a, b) in SHAPES + [(8, 4)]:
    w["trans_%d_%d" % (a, b)] = rw.TransitionWindow(np.ones(a + b), a, b)
    x = rng.normal(0, 0.3, a + b)
    w["x_%d_%d" % (a, b)] = x
    w["xwin_%d_%d" % (a, b)] = rw.TransitionWindow(x, a, b)
np.savez_compressed(os.path.join(HERE, "ref_window.npz"), **w)

# ------------------------------------------------------------------------------------------------ mdct
m = {}
for (a, b) in SHAPES + [(8, 4), (4, 4)]:
    x = pcm_to_float(gauss_pcm(2 * (a + b), 0.2)).reshape(2, a + b) if a >= 128 else rng.normal(0, 1, (2, a + b))
    m["x_%d_%d" % (a, b)] = x
    m["mdct_%d_%d" % (a, b)] = np.array([rmdct.MDCT(r, a, b) for r in x])
    m["slow_%d_%d" % (a, b)] = np.array([rmdct.MDCTslow(r, a, b) for r in x]) if a + b <= 256 else np.zeros(0)
    X = rng.normal(0, 1e-2, (2, (a + b) // 2))
    m["X_%d_%d" % (a, b)] = X
    m["imdct_%d_%d" % (a, b)] = np.array([rmdct.IMDCT(r, a, b) for r in X])
np.savez_compressed(os.path.join(HERE, "ref_mdct.npz"), **m)


# ------------------------------------------------------------------------------------------------ smr
def test_blocks(n):
    """three kinds of 16-bit content of length n: noise, tones on a noise floor, quiet"""
    t = np.arange(n)
    noise = gauss_pcm(n, 0.1)
    tones = np.clip(np.rint(9000 * np.sin(2 * np.pi * 0.021 * t) + 4000 * np.sin(2 * np.pi * 0.13 * t + 1)
                            + 2000 * np.sin(2 * np.pi * 0.31 * t + 2)) + gauss_pcm(n, 0.003), -32767, 32767).astype(np.int16)
    quiet = gauss_pcm(n, 0.0004)
    return np.array([noise, tones, quiet])


s = {}
for fs in (48000, 44100):
    for (a, b) in SHAPES:
        N = a + b
        sfb = bands(a, b, fs)
        pcm = test_blocks(N)
        thr, smr, scl = [], [], []
        for row in pcm:
            x = pcm_to_float(row)
            X = rmdct.MDCT(rw.TransitionWindow(x, a, b), a, b)[:N // 2]
            sc = rq.ScaleFactor(np.max(np.abs(X)), 4)
            X = X * (1 << sc)
            thr.append(rp.getMaskedThreshold(x, X, sc, fs, sfb))
            smr.append(rp.CalcSMRs(x, X, sc, fs, sfb))
            scl.append(sc)
        key = "%d_%d_%d" % (a, b, fs)
        s["pcm_" + key], s["thr_" + key], s["smr_" + key], s["scale_" + key] = pcm, np.array(thr), np.array(smr), np.array(scl)
np.savez_compressed(os.path.join(HERE, "ref_smr.npz"), **s)


# ------------------------------------------------------------------------------------------------ encode chains
def params(fs, nch, n_scale=4, n_mant=4, tbps=2.86):
    cp = types.SimpleNamespace()           # audiofile.py:51-53 is an empty attribute bag
    cp.sampleRate, cp.nChannels, cp.nMDCTLines = fs, nch, 1024
    cp.nScaleBits, cp.nMantSizeBits, cp.targetBitsPerSample = n_scale, n_mant, tbps
    cp.nSamplesPerBlock, cp.bitReservoir, cp.nSamplesShort = 1024, 0, 128
    cp.a = cp.b = 1024
    cp.blkswBitA = cp.blkswBitB = 1
    return cp


def shape_cycle(n_hops, transient_hops):
    """block shapes as the reference CLI produces them (pacfileThem.py:1192-1210): a transient hop becomes
    eight 128-sample blocks; a <- b after every block."""
    shapes, a = [], 1024
    for h in range(n_hops):
        if h in transient_hops:
            for _ in range(8):
                shapes.append((a, 128)); a = 128
        else:
            shapes.append((a, 1024)); a = 1024
    return shapes


def stereo_pcm(n_hops):
    n = n_hops * 1024
    g1, g2 = gauss_pcm(n, 0.1).astype(np.float64), gauss_pcm(n, 0.1).astype(np.float64)
    hop = np.arange(n) // 1024
    right = np.where(hop % 2 == 0, 0.8 * g1 + 0.2 * g2, 0.1 * g2)
    lvl = 10.0 ** (-1.5 * (hop % 5 == 3))                      # quieter hops: short mantissas, Huffman wins
    pcm = np.stack([g1 * lvl, right * lvl])
    return np.clip(np.rint(pcm), -32767, 32767).astype(np.int16)


def run_chain(fn_name, pcm, shapes, cp, joint):
    """Feed consecutive blocks (prior samples + new samples) to a reference encode function the way
    WriteDataBlock / JointWriteDataBlock do (pacfileThem.py:628-645, 799-816); returns per-block results."""
    x = np.array([pcm_to_float(ch) for ch in pcm])
    nch = x.shape[0]
    prior = np.zeros((nch, 1024))
    pos, out = 0, []
    fn = getattr(rc, fn_name)
    for (a, b) in shapes:
        new = x[:, pos:pos + b]
        pos += b
        full = [np.concatenate((prior[c][-a:], new[c])) for c in range(nch)]
        prior = new
        cp.a, cp.b = a, b
        cp.sfBands = bands(a, b, cp.sampleRate)
        res_in = cp.bitReservoir
        r = fn([f.copy() for f in full], cp)
        out.append((a, b, res_in, cp.bitReservoir, r))
    return out


def dense(m, ba, nlines):
    """compact int mantissa array -> dense [half] plane (0 where the band has no bits)"""
    out = np.zeros(int(np.sum(nlines)), dtype=np.int64)
    lo = np.cumsum(nlines) - nlines
    i = 0
    for k in range(len(nlines)):
        if ba[k]:
            out[lo[k]:lo[k] + nlines[k]] = m[i:i + nlines[k]]
            i += nlines[k]
    assert i == len(m)
    return out


def code_strings(m):
    """Huffman-coded mantissa list (codecThem.py:190-200) as one newline-joined string"""
    return "\n".join(str(v) for v in m)


def store_chain(e, tag, pcm, chain, cp, joint, huff):
    e[tag + "_pcm"] = pcm
    e[tag + "_shapes"] = np.array([(a, b) for (a, b, *_r) in chain])
    e[tag + "_res_in"] = np.array([c[2] for c in chain])
    e[tag + "_res_out"] = np.array([c[3] for c in chain])
    e[tag + "_params"] = np.array([cp.sampleRate, cp.nChannels, cp.nScaleBits, cp.nMantSizeBits, cp.targetBitsPerSample])
    for i, (a, b, _ri, _ro, r) in enumerate(chain):
        nl = bands(a, b, cp.sampleRate).nLines
        k = "%s_%d" % (tag, i)
        if joint:
            sf, ba, mant, osf, ms = r[0], r[1], r[2], r[3], r[4]
            tbl = r[5] if huff else [15, 15]
            e[k + "_ms"] = np.array(ms, dtype=np.int64)
        else:
            sf, ba, mant, osf = r[0], r[1], r[2], r[3]
            tbl = r[4]
        e[k + "_sf"] = np.array(sf, dtype=np.int64)
        e[k + "_ba"] = np.array(ba, dtype=np.int64)
        e[k + "_os"] = np.array(osf, dtype=np.int64)
        e[k + "_table"] = np.array(tbl, dtype=np.int64)
        for c in range(len(sf)):
            if tbl[c] == 15:
                e[k + "_mant%d" % c] = dense(np.asarray(mant[c]), np.asarray(ba[c]), nl)
            else:
                e[k + "_codes%d" % c] = np.array(code_strings(mant[c]))


e = {}
e["pcmmap_in"] = np.array([-32768, -32767, -32766, -12345, -2, -1, 0, 1, 2, 777, 32766, 32767], dtype=np.int16)
e["pcmmap_out"] = pcm_to_float(e["pcmmap_in"])            # pcmfile.py:91-100 (-32768 -> -0.0/0.0)
cwd = os.getcwd()
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    try:
        # (1) no ./training_data: every table loop is empty -> raw mantissas (table id 15)
        pcm = np.array([gauss_pcm(7 * 1024, 0.1)])
        cp = params(48000, 1)
        shapes = shape_cycle(7, {3})
        # EncodeSingleChannel directly (mono chain; the caller carries the reservoir like Encode without savings)
        x = pcm_to_float(pcm[0]); prior = np.zeros(1024); pos = 0; chain = []
        for (a, b) in shapes:
            new = x[pos:pos + b]; pos += b
            full = np.concatenate((prior[-a:], new)); prior = new
            cp.a, cp.b, cp.sfBands = a, b, bands(a, b, 48000)
            ri = cp.bitReservoir
            sf, ba, mant, osf = rc.EncodeSingleChannel(full.copy(), cp)
            chain.append((a, b, ri, cp.bitReservoir, ([sf], [ba], [mant], [osf], [15])))
        store_chain(e, "single", pcm, chain, cp, False, False)
        # EncodeNoHuff with the training-script parameters (huffman_training_script.py:39-45,60), 2 channels, 44.1 kHz
        pcm = stereo_pcm(6)
        cp = params(44100, 2, n_scale=3, n_mant=5, tbps=2.27)
        chain = run_chain("EncodeNoHuff", pcm, shape_cycle(6, {2}), cp, False)
        store_chain(e, "nohuff", pcm, chain, cp, False, False)
        # JointEncodeChannels, no tables
        pcm = stereo_pcm(8)
        cp = params(48000, 2)
        x = np.array([pcm_to_float(c) for c in pcm]); prior = np.zeros((2, 1024)); pos = 0; chain = []
        for (a, b) in shape_cycle(8, {4}):
            new = x[:, pos:pos + b]; pos += b
            full = [np.concatenate((prior[c][-a:], new[c])) for c in range(2)]; prior = new
            cp.a, cp.b, cp.sfBands = a, b, bands(a, b, 48000)
            ri = cp.bitReservoir
            r = rc.JointEncodeChannels(full[0].copy(), full[1].copy(), cp)
            chain.append((a, b, ri, cp.bitReservoir, r))
        store_chain(e, "jointch", pcm, chain, cp, True, False)

        # (2) with tables: written here from this repo's table data, one directory per table so that
        # os.walk/glob finds them; the order found is recorded and decides the table ids of this fixture
        # the reference numbers tables in os.walk/glob order (filesystem dependent): create the files,
        # see which order they are found in, THEN write table TABLE_ORDER[i] into the i-th file found, so that the
        # reference iterates the tables in this repo's fixed order (ties between tables go to the first one)
        for i in range(len(HT.TABLE_ORDER)):
            os.makedirs(os.path.join("training_data", "t%d" % i))
            open(os.path.join("training_data", "t%d" % i, "t%d_table.pkl" % i), "wb").close()
        found = [y for x in os.walk("./training_data/") for y in rc.glob(os.path.join(x[0], "*table.pkl"))]
        assert len(found) == len(HT.TABLE_ORDER)
        for path, name in zip(found, HT.TABLE_ORDER):
            table, escape = HT.TABLES[name]
            d = H.Py2Dict((int(v), (str(code), int(n))) for v, (code, n) in table.items())
            with open(path, "wb") as f:
                pickle.dump((d, int(escape)), f, protocol=2)
        e["table_order"] = np.array(HT.TABLE_ORDER)
        pcm = stereo_pcm(10)
        cp = params(48000, 2)
        chain = run_chain("JointEncode", pcm, shape_cycle(10, {3, 7}), cp, True)
        store_chain(e, "joint", pcm, chain, cp, True, True)
        pcm = stereo_pcm(6)
        cp = params(48000, 2)
        chain = run_chain("Encode", pcm, shape_cycle(6, {2}), cp, False)
        store_chain(e, "indep", pcm, chain, cp, False, True)
        # low rate: small mantissas, the Huffman tables win often (and the reservoir grows)
        pcm = (stereo_pcm(8).astype(np.float64) * 0.05).astype(np.int16)
        cp = params(48000, 2, tbps=1.4)
        chain = run_chain("JointEncode", pcm, shape_cycle(8, {5}), cp, True)
        store_chain(e, "jointlo", pcm, chain, cp, True, True)
        # the parameters the tables were trained with (huffman_training_script.py:39-45), tonal + quiet content
        t = np.arange(8 * 1024)
        tone = 6000 * np.sin(2 * np.pi * 440.0 / 44100 * t) + 2500 * np.sin(2 * np.pi * 1320.0 / 44100 * t + 0.5)
        pcm = np.clip(np.rint(np.stack([tone, 0.7 * tone]) + gauss_pcm(2 * 8 * 1024, 0.002).reshape(2, -1)),
                      -32767, 32767).astype(np.int16)
        cp = params(44100, 2, n_scale=3, n_mant=5, tbps=2.27)
        chain = run_chain("JointEncode", pcm, shape_cycle(8, {6}), cp, True)
        store_chain(e, "jointtrain", pcm, chain, cp, True, True)

        # (3) decode side: Decode / JointDecode of raw-mantissa blocks (codecThem.py:30-134)
        pcm = stereo_pcm(5)
        cp = params(48000, 2)
        x = np.array([pcm_to_float(c) for c in pcm]); prior = np.zeros((2, 1024)); pos = 0
        for i, (a, b) in enumerate(shape_cycle(5, {2})):
            new = x[:, pos:pos + b]; pos += b
            full = [np.concatenate((prior[c][-a:], new[c])) for c in range(2)]; prior = new
            cp.a, cp.b, cp.sfBands = a, b, bands(a, b, 48000)
            nl = cp.sfBands.nLines
            sf, ba, mant, osf, ms = rc.JointEncodeChannels(full[0].copy(), full[1].copy(), cp)
            d0, d1 = dense(mant[0], ba[0], nl), dense(mant[1], ba[1], nl)      # the decoders index DENSE planes
            out = rc.JointDecode(sf, ba, [d0.copy(), d1.copy()], osf, cp, ms)
            k = "dec_%d" % i
            e[k + "_shape"] = np.array([a, b])
            e[k + "_sf"], e[k + "_ba"], e[k + "_os"], e[k + "_ms"] = np.array(sf), np.array(ba), np.array(osf), np.array(ms)
            e[k + "_mant0"], e[k + "_mant1"] = d0, d1
            e[k + "_jointdec"] = np.array(out)
            cp1 = copy.copy(cp)
            s1, b1, m1, o1 = rc.EncodeSingleChannel(full[0].copy(), cp1)
            dm = dense(m1, b1, nl)
            e[k + "_sf1"], e[k + "_ba1"], e[k + "_os1"], e[k + "_mant1ch"] = np.array(s1), np.array(b1), np.array(o1), dm
            e[k + "_dec"] = np.array(rc.Decode(s1, b1, dm.copy(), o1, cp1))     # per channel (codecThem.py:30)
        e["dec_pcm"] = pcm
        e["dec_n"] = np.array(i + 1)
    finally:
        os.chdir(cwd)
np.savez_compressed(os.path.join(HERE, "ref_encode.npz"), **e)
for f in ("ref_psychoac", "ref_window", "ref_mdct", "ref_smr", "ref_encode"):
    print(f, os.path.getsize(os.path.join(HERE, f + ".npz")) // 1024, "KiB")
