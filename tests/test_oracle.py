"""
Pins the oracle (CPU restatement) against
  * golden vectors recorded from the reference's py3-importable modules (tests/golden/*.npz,
    generator tests/golden/make_golden.py),
  * the reference's own stated relations (mdct.py:131-199),
  * property checks where the reference offers nothing (KBD window),
and checks the two oracle flavours (faithful vs fast) against each other.  CPU only.
"""
import math
import os

import numpy as np
import pytest

from oracle import bitalloc, codec, fast, mdct, ms_stereo, psychoac, quantize, window


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


# ------------------------------------------------------------------ golden: quantize.py
def test_scale_factor_golden(golden_dir):
    g = _load(golden_dir, "quantize.npz")
    for (s, m), want in zip(g["sf_cases"], g["sf_out"]):
        got = [quantize.ScaleFactor(float(v), int(s), int(m)) for v in g["vals"]]
        assert got == list(want)
        # the batched form used by oracle.fast / mirrored by the HIP kernel
        nonneg = np.abs(g["vals"])
        assert (fast.scale_factor_batch(nonneg, int(s), int(m)) ==
                [quantize.ScaleFactor(float(v), int(s), int(m)) for v in nonneg]).all()


def test_scale_factor_ba0_cap_is_13():
    # nMantBits = 0 -> 15-bit code -> at most 13 leading zeros, never 15
    assert quantize.ScaleFactor(0.0, 4, 0) == 13
    assert quantize.ScaleFactor(0.0, 4, 2) == 15


def test_quantize_uniform_golden(golden_dir):
    g = _load(golden_dir, "quantize.npz")
    for nb, wantv, wants in zip(g["nbits_cases"], g["vquant_out"], g["quant_out"]):
        assert np.array_equal(quantize.vQuantizeUniform(g["vals"], int(nb)), wantv)
        assert [quantize.QuantizeUniform(float(v), int(nb)) for v in g["vals"]] == list(wants)


def test_mantissa_golden(golden_dir):
    g = _load(golden_dir, "quantize.npz")
    for (sc, sb, mb), wantv, wants in zip(g["mant_cases"], g["vmant_out"], g["mant_out"]):
        got = quantize.vMantissa(g["vals"], int(sc), int(sb), int(mb))
        assert np.array_equal(got, wantv)
        assert [quantize.Mantissa(float(v), int(sc), int(sb), int(mb)) for v in g["vals"][:120]] == list(wants)
        assert np.array_equal(fast.mantissa_batch(g["vals"], int(sc), int(sb), int(mb)), wantv.astype(np.int64))


def test_pcm_to_float_golden(golden_dir):
    g = _load(golden_dir, "quantize.npz")
    assert np.array_equal(quantize.vDequantizeUniform(g["codes16"], 16), g["vdequant16_out"])
    # SURVEY A.4 probes
    assert quantize.vDequantizeUniform(np.array([1.0]), 16)[0] == 3.0518043793392844e-05
    assert quantize.vDequantizeUniform(np.array([32768.0]), 16)[0] == 0.0


def test_floor_log2_matches_libm_quotient():
    # quantize.py:136 uses int(math.log(m, 2)); the batched oracle and the HIP kernel use an exact
    # integer floor(log2).  They agree on every power of two and its neighbours in range.
    for k in range(0, 31):
        for m in (2 ** k - 1, 2 ** k, 2 ** k + 1):
            if m > 0:
                assert int(math.log(m, 2)) == m.bit_length() - 1
    codes = np.random.default_rng(1).integers(1, 2 ** 31, 200000)
    assert (fast._floor_log2(codes) == np.array([int(math.log(int(c), 2)) for c in codes])).all()


# ------------------------------------------------------------------ golden: bitalloc.py
def test_bitalloc_golden(golden_dir):
    g = _load(golden_dir, "bitalloc.npz")
    for i in range(int(g["n"])):
        smr = g["smr_%d" % i].copy()
        bits, left = bitalloc.BitAlloc(float(g["budget_%d" % i]), int(g["maxb_%d" % i]), len(smr),
                                       g["nlines_%d" % i], smr)
        assert np.array_equal(bits, g["bits_%d" % i]), i
        assert left == int(g["left_%d" % i]), i
        assert np.array_equal(smr, g["smr_after_%d" % i]), i      # SMR is mutated in place


def test_bitalloc_negative_remainder():
    bits, left = bitalloc.BitAlloc(5, 16, 2, np.array([4, 4]), np.array([10., 5.]))
    assert list(bits) == [2., 0.] and left == -3


# ------------------------------------------------------------------ golden: ms_stereo.py
def test_ms_stereo_golden(golden_dir):
    g = _load(golden_dir, "ms_stereo.npz")
    for k in range(int(g["n"])):
        sfb = psychoac.ScaleFactorBands(g["nlines_%d" % k])
        L, R = g["L_%d" % k], g["R_%d" % k]
        sw = ms_stereo.MSSwitchSFBands(L, R, sfb)
        assert sw == list(g["switch_%d" % k])
        assert np.array_equal(fast.ms_switch_batch(L[None], R[None], sfb)[0], g["switch_%d" % k])
        smf = ms_stereo.StereoMaskingFactor(g["midT_%d" % k], g["sideT_%d" % k], sfb, g["z_%d" % k])
        assert np.array_equal(smf[0], g["smf0_%d" % k]) and np.array_equal(smf[1], g["smf1_%d" % k])
        s = g["smrs_%d" % k]
        o1, o2 = ms_stereo.OverallSMRs(s[0], s[1], s[2], s[3], sfb, sw)
        assert np.array_equal(o1, g["o1_%d" % k]) and np.array_equal(o2, g["o2_%d" % k])


# ------------------------------------------------------------------ golden: window.py (Hann)
def test_hann_golden(golden_dir):
    g = _load(golden_dir, "window.npz")
    for N in (2048, 1152, 256, 8):
        assert np.array_equal(window.HanningWindow(g["x_%d" % N]), g["hann_%d" % N])


# ------------------------------------------------------------------ KBD: property checks (the table itself is pinned in test_reference_golden.py)
@pytest.mark.parametrize("N", [256, 2048])
def test_kbd_princen_bradley(N):
    w = window.kbd_table(N)
    assert np.allclose(w[:N // 2] ** 2 + w[N // 2:] ** 2, 1.0, atol=1e-14)
    assert np.allclose(w, w[::-1], atol=1e-14)
    assert w[0] < 1e-4 and abs(w[N // 2 - 1] - 1) < 1e-6 and np.all(np.diff(w[:N // 2]) > 0)
    # closed form of Bosi & Goldberg pp.108-109 (the formula window.py:52-53 cites)
    M = N // 2
    j = np.arange(M + 1)
    k = np.i0(np.pi * 4.0 * np.sqrt(1 - ((j - M / 2) / (M / 2)) ** 2)) ** 2
    ref = np.sqrt(np.cumsum(k)[:M] / np.sum(k))
    assert np.allclose(w[:M], ref, rtol=1e-12)


def test_transition_window_shapes():
    x = np.random.default_rng(0).normal(size=1152)
    y = window.TransitionWindow(x, 1024, 128)
    assert np.array_equal(y[:1024], (x * np.append(window.kbd_table(2048)[:1024], np.zeros(128)))[:1024])
    assert np.array_equal(y[1024:], x[1024:] * window.kbd_table(256)[128:])
    assert np.array_equal(y, x * fast.transition_table(1024, 128))


# ------------------------------------------------------------------ MDCT: the reference's own relations
def test_mdct_equals_mdctslow_reference_case():
    x = np.arange(1024.0)                         # mdct.py:185-199
    np.testing.assert_array_almost_equal(mdct.MDCT(x, 512, 512), mdct.MDCTslow(x, 512, 512))


@pytest.mark.parametrize("ab", [(1024, 1024), (128, 128), (1024, 128), (128, 1024)])
def test_mdct_equals_definition_all_shapes(ab):
    a, b = ab
    x = np.random.default_rng(3).normal(0, 0.1, a + b)
    X = mdct.MDCT(x, a, b)
    assert np.abs(X - mdct.MDCTslow(x, a, b)).max() <= 1e-12 * max(1.0, np.abs(X).max()) * (a + b)


def test_tdac_known_answer():
    # mdct.py:131-182: un-windowed MDCTslow -> 0.5*inverse -> overlap-add reproduces x (after a 4-sample delay)
    x = np.array([3, 3, 3, 3, 2, 0, -2, -4, -1, 0, 1, 2], dtype=float)
    a = b = 4
    z = np.zeros(4)
    prev = np.zeros(8)
    out = np.zeros(0)
    nblocks = 4
    for nb in range(nblocks):
        cur = x[nb * 4:nb * 4 + 4] if nb < 3 else z
        prior = z if nb == 0 else x[(nb - 1) * 4:(nb - 1) * 4 + 4]
        blk = np.concatenate([prior, cur])
        im = 0.5 * mdct.MDCTslow(mdct.MDCTslow(blk, a, b), a, b, True)
        out = np.concatenate([out, im[:b] + prev[b:]])
        prev = im
    assert np.array_equal(np.rint(out[b:]).astype(int), x.astype(int))
    # first intermediate rows stated in the reference comments (mdct.py:174-177)
    first = 0.5 * mdct.MDCTslow(mdct.MDCTslow(np.concatenate([z, x[:4]]), a, b), a, b, True)
    np.testing.assert_array_almost_equal(first, [0, 0, 0, 0, 3, 3, 3, 3])


# ------------------------------------------------------------------ band tables
def test_band_tables_48k():
    assert list(fast.bands_for(1024, 1024).nLines) == [4, 5, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 12, 14, 16, 19, 24,
                                                       30, 38, 47, 56, 76, 107, 149, 363]
    assert list(fast.bands_for(128, 128).nLines) == [2, 1, 3, 3, 5, 9, 18, 42, 45]
    assert list(fast.bands_for(1024, 128).nLines) == [7, 8, 11, 15, 24, 41, 79, 187, 204]
    assert list(fast.bands_for(128, 1024).nLines) == [7, 8, 11, 15, 24, 41, 79, 187, 204]


def test_budgets_48k():
    P = fast.DEFAULTS
    assert abs(fast.mono_budget(P, 1024, 25) - 2722.64) < 1e-9
    assert abs(fast.mono_budget(P, 576, 9) - 1569.36) < 1e-9
    assert abs(fast.mono_budget(P, 128, 9) - 288.08) < 1e-9
    assert abs(fast.joint_budget(P, 1024, 25, 0) - 5414.28) < 1e-9
    assert abs(fast.joint_budget(P, 576, 9, 0) - 3123.72) < 1e-9
    assert abs(fast.joint_budget(P, 128, 9, 0) - 561.16) < 1e-9


# ------------------------------------------------------------------ faithful vs fast
def _pcm(seed, n, sigma=0.1):
    p = np.clip(np.rint(np.random.default_rng(seed).normal(0, sigma * 32767, n)), -32767, 32767)
    return np.sign(p) * 2.0 * np.abs(p) / 65535


def _check_mono(blocks, a, b, res_in):
    r = fast.encode_mono_batch(blocks, a, b, reservoir_in=res_in)
    for i in range(blocks.shape[0]):
        cp = codec.default_params()
        cp.a, cp.b = a, b
        cp.sfBands = codec.bands_for_block(a, b, 1024, 48000)
        cp.bitReservoir = int(res_in[i])
        sf, ba, m, o = codec.EncodeSingleChannel(blocks[i].copy(), cp)
        assert o == r["overall_scale"][i]
        assert np.array_equal(sf, r["scale_factor"][i]) and np.array_equal(ba, r["bit_alloc"][i])
        assert cp.bitReservoir == r["reservoir_out"][i]
        assert np.array_equal(m, fast.compact_mantissa(r["mantissa"][i], ba, cp.sfBands))


def test_fast_equals_faithful_mono_long():
    x = _pcm(1234, 5 * 1024)
    _check_mono(np.array(fast.blocks_from_stream(x, 1024)), 1024, 1024, [0, 17, -40, 300])


@pytest.mark.parametrize("ab", [(128, 128), (1024, 128), (128, 1024)])
def test_fast_equals_faithful_mono_other_shapes(ab):
    a, b = ab
    x = _pcm(42, 6000, 0.3)
    _check_mono(np.stack([x[s:s + a + b] for s in (0, 777, 2500)]), a, b, [0, 5, -3])


def test_fast_equals_faithful_joint():
    g1, g2 = _pcm(1234, 4 * 1024), _pcm(5678, 4 * 1024)
    R = g1.copy()
    for h in range(4):
        sl = slice(h * 1024, (h + 1) * 1024)
        R[sl] = 0.8 * g1[sl] + 0.2 * g2[sl] if h % 2 == 0 else 0.1 * g2[sl]
    bl = np.array(fast.blocks_from_stream(g1, 1024))
    br = np.array(fast.blocks_from_stream(R, 1024))
    r = fast.encode_joint_batch(bl, br, 1024, 1024, reservoir_in=[0, 11, -9])
    seen = set()
    for i in range(3):
        cp = codec.default_params(nChannels=2)
        cp.bitReservoir = [0, 11, -9][i]
        sf, ba, m, o, sw = codec.JointEncodeChannels(bl[i].copy(), br[i].copy(), cp)
        assert list(o) == list(r["overall_scale"][i]) and sw == list(r["ms_switch"][i])
        seen.update(sw)
        for c in range(2):
            assert np.array_equal(sf[c], r["scale_factor"][i, c]) and np.array_equal(ba[c], r["bit_alloc"][i, c])
            assert np.array_equal(m[c], fast.compact_mantissa(r["mantissa"][i, c], ba[c], cp.sfBands))
        assert cp.bitReservoir == r["reservoir_out"][i]
    assert seen == {0, 1}          # both branches of the M/S switch exercised


def test_silence_and_sine_blocks():
    # digital silence: no peaks, everything floors; 1 kHz sine: a handful of maskers
    sil = np.zeros((1, 2048))
    r = fast.encode_mono_batch(sil, 1024, 1024)
    assert r["overall_scale"][0] == 15 and not r["mantissa"].any()
    n = np.arange(3 * 1024)
    p = np.rint(0.5 * 32767 * np.sin(2 * np.pi * 1000 * n / 48000))
    x = np.sign(p) * 2.0 * np.abs(p) / 65535
    _check_mono(np.array(fast.blocks_from_stream(x, 1024)), 1024, 1024, [0, 0])


# ------------------------------------------------------------------ Huffman gain (host-side stage)
def test_huffman_gain_escape_undercount_and_order():
    cp = codec.default_params()
    ba = np.zeros(25, dtype=int); ba[0] = 4; ba[1] = 3
    m = np.array([0, 0, 4, 0, 2, 0, 0, 4, 0], dtype=np.int32)
    t, codes, saved = codec.calculateHuffmanGain(m, ba, cp)
    assert t == 0 and saved == (4 * 4 + 3 * 5) - (1 + 1 + 3 + 1 + 3 + 1 + 1 + 3 + 1)   # percussive wins the tie with tonal
    assert codes[2] == "101"
    m2 = np.array([16, 1, 1, 1, 1, 1, 1, 1, 1], dtype=np.int32)                       # 16 = percussive escape value
    assert codec.huffman_cost(m2, ba, cp.sfBands, *codec.TABLES["percussive"], 10 ** 9) == 6 + 8 * 4


# ------------------------------------------------------------------ bit packer / .pac framing
def test_bitpack_known_answer():
    # bitpack.py:183-196: x = (3,5,11,3,1) in (4,3,5,3,1) bits -> 0011 101 0|1011 011 1
    from oracle.pacfile import PackedBits
    pb = PackedBits()
    pb.Size(2)
    for v, n in zip((3, 5, 11, 3, 1), (4, 3, 5, 3, 1)):
        pb.WriteBits(v, n)
    assert pb.GetPackedData() == bytes([0x3A, 0xB7])
    pb.Size(3)
    pb.WriteBits(0x1ABCD, 17)           # spans three bytes, only the low 17 bits of info are written
    pb.WriteBits(0xFF, 0)
    assert pb.GetPackedData() == bytes([0xD5, 0xE6, 0x80])


def test_pac_stream_structure():
    # header + one chunk pair per block + the non-joint flush block; every chunk length adds up
    import struct
    from mrcaudiocodec_amd import synth
    from oracle import pacfile
    s = synth.c3_stereo(3)
    shapes = [(i * 1024, 1024, 1024) for i in range(3)]
    data = pacfile.encode_stereo_stream(s, shapes)
    assert data[:4] == b"PAC "
    sr, nch, nsamp, nlines, sb, mb = struct.unpack("<LHLLHH", data[4:22])
    assert (sr, nch, nlines, sb, mb) == (48000, 2, 1024, 4, 4) and nsamp == 3 * 1024 + 1024
    nb, = struct.unpack("<L", data[22:26])
    assert nb == 25
    pos = 26 + 2 * nb
    chunks = 0
    while pos < len(data):
        n, = struct.unpack("<L", data[pos:pos + 4])
        pos += 4 + n
        chunks += 1
    assert pos == len(data) and chunks == 2 * (3 + 1)


# ------------------------------------------------------------------ the chained encode's bit allocation, as a model
def _alloc_by_event_list(budget, max_mant, n_lines, smr):
    """The scheme of csrc/mrc_kernels_chain.hip restated in plain Python: the grant ATTEMPTS of bitalloc.py:106-155 in the
    order np.argmax serves them (keys S, S-12, S-18, ... per band, ties by band), their cost prefix sums, a cut where fewer
    than max(nLines) bits would be left, then the tail event by event with immediate retirement of bands that no longer
    fit; integer budget tests (nLines <= left <=> nLines + spent <= floor(budget); left > 0 <=> spent < ceil(budget))."""
    import math
    n_tot, K = len(smr), max_mant - 1
    evs = []
    for i in range(n_tot):
        cur = float(smr[i])
        for k in range(K):
            evs.append((cur, i, k))
            cur = cur - 12.0 if k == 0 else cur - 6.0
    evs.sort(key=lambda t: (-t[0], t[1]))
    pre = [0]
    for (_, i, k) in evs:
        pre.append(pre[-1] + (2 * n_lines[i] if k == 0 else n_lines[i]))
    bf, bc = math.floor(budget), math.ceil(budget)
    max_n = max(n_lines)
    cut = sum(1 for p in range(len(evs)) if pre[p] + max_n <= bf)
    bits = [0] * n_tot
    for p in range(cut):
        bits[evs[p][1]] = evs[p][2] + 2
    spent = pre[cut]
    alive = {i for i in range(n_tot) if n_lines[i] + spent <= bf}
    e = cut
    while e < len(evs) and spent < bc and alive:
        _, i, k = evs[e]
        e += 1
        if i not in alive:
            continue
        spent += 2 * n_lines[i] if k == 0 else n_lines[i]       # (a live band always fits)
        bits[i] = k + 2
        alive = {j for j in alive if n_lines[j] + spent <= bf}
    return np.array(bits), int(budget - spent)


def test_bitalloc_event_list_model_equals_greedy_loop():
    # what chain_prep_kernel / chain_phase_b_kernel implement on the device, against the oracle's restatement of
    # bitalloc.py:106-155: ties, cross-level ties (multiples of 6), equal SMRs, budgets <= 0, huge budgets (the 16-bit
    # cap), every maxMantBits, the band tables of all block shapes
    from oracle.bitalloc import BitAlloc
    rng = np.random.default_rng(1)
    nl25 = np.array([4, 5, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 12, 14, 16, 19, 24, 30, 38, 47, 56, 76, 107, 149, 363])
    nl9 = np.array([2, 1, 3, 3, 5, 9, 18, 42, 45])
    for trial in range(700):
        kind = trial % 6
        nl = (nl25, np.concatenate([nl25, nl25]), nl9, np.concatenate([nl9, nl9]))[kind] if kind < 4 \
            else rng.integers(1, 50, size=rng.integers(1, 30))
        smr = rng.normal(0, 20, len(nl))
        if trial % 7 == 0:
            smr = np.round(smr)
        if trial % 11 == 0:
            smr[:] = smr[0]
        if trial % 13 == 0:
            smr = np.round(smr / 6) * 6 + 0.0
        budget = float(rng.choice([5414.28, 2722.64, 561.16, 288.08, 5.0, -3.2, 100000.5, rng.uniform(0, 9000),
                                   float(rng.integers(0, 6000))]))
        max_m = int(rng.choice([16, 16, 16, 8, 4, 2]))
        b0, l0 = BitAlloc(budget, max_m, len(nl), nl, smr.copy())
        b1, l1 = _alloc_by_event_list(budget, max_m, [int(v) for v in nl], smr)
        assert np.array_equal(np.asarray(b0).astype(int), b1) and int(l0) == l1, (trial, budget, max_m)
