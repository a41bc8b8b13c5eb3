"""
GPU tests of the sensitivity certificate (mrc_get_sensitivity, MRC_OPT_SENSITIVITY): the encode calls count the integer
decisions they took within a guard band of floating-point rounding -- quantiser / scale-factor edges (quantize.py:12-38,
114-146, 294-322), ties of the greedy bit allocation (bitalloc.py:132-151), the M/S test's 0.8 threshold (ms_stereo.py:5-27),
the strict peak test (psychoac.py:162).  An ordinary corpus reports none; crafted near-ties are flagged; with the guards a
10^8 times wider (option value 2) the counts equal a NumPy model evaluated on the ORACLE's intermediate values.
"""
import numpy as np
import pytest

from oracle import fast

pytestmark = pytest.mark.gpu
SENS = 5                                                     # MRC_OPT_SENSITIVITY


@pytest.fixture()
def h():
    from mrcaudiocodec_amd import Handle
    hd = Handle(device_id=0)
    yield hd
    hd.close()


def _blocks(x, n, hop=1024):
    return np.array(fast.blocks_from_stream(x, hop, n))


def test_an_ordinary_corpus_reports_no_decision_near_an_edge(h):
    from mrcaudiocodec_amd import synth, pacfile as ppac
    h.set_option(SENS, 1)
    h.sensitivity()
    n = 192
    h.encode_mono(_blocks(synth.c2_noise(n + 1), n), 1024, 1024)
    xs = synth.c3_stereo(65)
    h.encode_joint(_blocks(xs[0], 64), _blocks(xs[1], 64), 1024, 1024)
    x, shapes = synth.c4_transients(24)
    stream = np.stack([x, 0.6 * x + 0.1 * synth.c2_noise(24, seed=2)])
    ppac.encode_stereo_stream(h, stream, shapes)              # the chained call: its scan's decisions are counted too
    s = h.sensitivity()
    assert s["blocks_examined"] == n + 64 + len(shapes) + 2
    for k in ("quantiser_edges", "bitalloc_near_ties", "ms_switch_near_threshold", "peak_near_ties"):
        assert s[k] == 0, s
    # the counters only run when asked for
    h.set_option(SENS, 0)
    h.encode_mono(_blocks(synth.c2_noise(9), 8), 1024, 1024)
    assert h.sensitivity()["blocks_examined"] == 0


def test_crafted_near_ties_are_flagged(h):
    from mrcaudiocodec_amd import synth
    h.set_option(SENS, 1)
    h.sensitivity()
    # M/S: R = L / 3 puts every band exactly on the threshold: |L^2 - R^2| = 8/9 L^2 = 0.8 (L^2 + R^2)
    bl = _blocks(synth.c2_noise(5), 4)
    h.encode_joint(bl, bl / 3.0, 1024, 1024)
    s = h.sensitivity()
    assert s["ms_switch_near_threshold"] >= 4 * 20, s
    # peak test: a block whose WINDOWED spectrum has two equal neighbouring bins -- built backwards: the inverse transform of
    # a spectrum with bins 100 and 101 equal, divided by the Hann window the analysis multiplies with (window.py:28-45)
    n = np.arange(2048)
    spec = np.zeros(1025, dtype=complex)
    spec[100] = spec[101] = 1.0
    tone = np.fft.irfft(spec, 2048) / (0.5 - 0.5 * np.cos(2 * np.pi * (n + 0.5) / 2048))
    tone *= 0.5 / np.abs(tone).max()
    h.encode_mono(tone[None, :], 1024, 1024)
    s = h.sensitivity()
    assert s["peak_near_ties"] >= 1, s
    assert s["blocks_examined"] == 1


def _model(ref, joint, guard_scale, n_scale_bits=4):
    """the device's QUANT and BITALLOC counts from the oracle's lines / SMRs / allocation (NumPy)"""
    sfb = ref["sfBands"]
    nb = sfb.nBands
    band_of_line = np.repeat(np.arange(nb), sfb.nLines)
    cap = (1 << n_scale_bits) - 1
    X = ref["mdct"] if joint else ref["mdct"][:, None, :]
    osc = ref["overall_scale"] if joint else ref["overall_scale"][:, None]
    smr = ref["smr"] if joint else ref["smr"][:, None, :]
    ba = ref["bit_alloc"] if joint else ref["bit_alloc"][:, None, :]
    sf = ref["scale_factor"] if joint else ref["scale_factor"][:, None, :]
    B = X.shape[0]
    quant = ties = 0
    g = 4e-13 * guard_scale
    for f in range(B):
        keys = []
        for strm in range(2 if joint else 1):
            sig = (np.where(ref["ms_switch"][f] == 1, 2 + strm, strm) if joint else np.zeros(nb, dtype=int))
            xs = np.abs(X[f][sig[band_of_line], np.arange(len(band_of_line))] * 2.0 ** osc[f][sig[band_of_line]])
            bl, sl = ba[f, strm][band_of_line], sf[f, strm][band_of_line]
            coded = (bl > 0) & (xs < 1.0)
            m = 2.0 ** (cap + bl) - 1
            t = (m * xs + 1.0) / 2.0
            step = 2.0 ** np.maximum(cap - sl, 0)
            q = t / step
            dist = np.abs(q - np.rint(q)) * step
            quant += int(np.count_nonzero(coded & (dist <= 0.5 * m * g)))
            for b in range(nb):
                if ba[f, strm, b] > 0:
                    peak = xs[band_of_line == b].max()
                    if 0.0 < peak < 1.0:
                        mb = 2.0 ** (cap + ba[f, strm, b]) - 1
                        tb = (mb * peak + 1.0) / 2.0
                        if tb >= 1.0:
                            e = np.floor(np.log2(tb))
                            if min(tb - 2.0 ** e, 2.0 ** (e + 1) - tb) <= 0.5 * mb * g:
                                quant += 1
            keys.append(smr[f][sig, np.arange(nb)])
        k = np.concatenate(keys)
        d = k[:, None] - k[None, :]
        r = d - 6.0 * np.rint(d / 6.0)
        near = (np.abs(r) <= 1e-9 * guard_scale) & (np.abs(d) < 1e6)
        ties += int(np.count_nonzero(np.triu(near, 1)))
    return quant, ties


@pytest.mark.parametrize("joint", [False, True])
def test_loose_guards_count_what_a_numpy_model_counts_on_the_oracle(h, joint):
    from mrcaudiocodec_amd import synth
    h.set_option(SENS, 2)                                     # every guard band x 1e8
    h.sensitivity()
    n = 32
    if joint:
        xs = synth.c3_stereo(n + 1)
        bl, br = _blocks(xs[0], n), _blocks(xs[1], n)
        h.encode_joint(bl, br, 1024, 1024)
        ref = fast.encode_joint_batch(bl, br, 1024, 1024)
    else:
        bl = _blocks(synth.c2_noise(n + 1), n)
        h.encode_mono(bl, 1024, 1024)
        ref = fast.encode_mono_batch(bl, 1024, 1024)
    s = h.sensitivity()
    quant, ties = _model(ref, joint, 1e8)
    assert quant > 5 and ties > 20                            # (the loose guards do catch something on this corpus)
    assert s["quantiser_edges"] == quant, (s, quant)
    assert s["bitalloc_near_ties"] == ties, (s, ties)
    assert s["blocks_examined"] == n


def test_cli_certificate(tmp_path):
    from mrcaudiocodec_amd import cli
    rng = np.random.default_rng(3)
    pcm = np.clip(np.rint(rng.normal(0, 0.05 * 32767, (2, 14 * 1024))), -32767, 32767).astype(np.int16)
    pcm[:, 6 * 1024:6 * 1024 + 128] = np.clip(np.rint(rng.normal(0, 0.5 * 32767, (2, 128))), -32767, 32767)   # a transient
    wav = tmp_path / "in.wav"
    wav.write_bytes(cli.wav_bytes(pcm, 48000))
    cert = {}
    data = cli.encode_wav(str(wav), certify=cert)
    assert data == cli.encode_wav(str(wav))                   # counting changes nothing
    assert cert["blocks_examined"] > 14 and cert["decisions_near_an_edge"] == 0
    assert "bytes_equal_exact_spread" not in cert
