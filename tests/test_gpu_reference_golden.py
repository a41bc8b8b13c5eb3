"""
GPU parity against outputs of the reference's OWN functions (tests/golden/ref_*.npz; see tests/refgold.py and
tests/golden/py2harness.py): the HIP path, called through the C ABI and through the drop-in module
mrcaudiocodec_amd.codecThem, must reproduce every integer the reference's EncodeSingleChannel /
JointEncodeChannels / Encode / EncodeNoHuff / JointEncode produced (scale factors, bit allocations, mantissas,
M/S switches, overall scales, Huffman table ids and code strings, reservoir after every block), the MDCT lines to
1e-12 of the block peak, thresholds / SMRs to 1e-9 dB and decoded blocks to 1e-12 of the block peak.
No oracle is involved in these comparisons: fixture in, fixture out.
"""
import numpy as np
import pytest

import refgold as G

pytestmark = pytest.mark.gpu

MDCT_RTOL = 1e-12
MRC_OPT_EXACT_SPREAD = 1          # include/mrc_hip.h
DB_ATOL = 1e-9


@pytest.fixture(scope="module")
def handles():
    from mrcaudiocodec_amd import Handle
    made = {}

    def get(fs=48000, **kw):
        key = (fs,) + tuple(sorted(kw.items()))
        if key not in made:
            made[key] = Handle(sample_rate=fs, device_id=0, **kw)
        return made[key]
    yield get
    for h in made.values():
        h.close()


def test_band_tables_on_handle_match_reference(handles):
    p = G.load("ref_psychoac.npz")
    for fs in (48000, 44100):
        h = handles(fs)
        for (a, b), key in (((1024, 1024), "1024_%d_cb"), ((1024, 128), "576_%d_short"), ((128, 128), "128_%d_short")):
            assert np.array_equal(h.bands(a, b), p["bt_nlines_" + key % fs])


def test_window_matches_reference(handles):
    w = G.load("ref_window.npz")
    h = handles()
    for (a, b) in G.SHAPES:
        got = h.window(np.ones((1, a + b)), a, b)[0]
        assert np.abs(got - w["trans_%d_%d" % (a, b)]).max() <= 4e-16, (a, b)
        got = h.window(w["x_%d_%d" % (a, b)][None], a, b)[0]
        assert np.abs(got - w["xwin_%d_%d" % (a, b)]).max() <= 1e-15, (a, b)


def test_mdct_matches_reference(handles):
    m = G.load("ref_mdct.npz")
    h = handles()
    for (a, b) in G.SHAPES:
        key = "%d_%d" % (a, b)
        lines, _ = h.mdct(m["x_" + key], a, b, apply_window=False)
        want = m["mdct_" + key]
        assert np.abs(lines - want).max() <= MDCT_RTOL * np.abs(want).max(), key


@pytest.mark.parametrize("fs", [48000, 44100])
@pytest.mark.parametrize("exact", [0, 1])
def test_threshold_and_smr_match_reference(handles, fs, exact):
    s = G.load("ref_smr.npz")
    h = handles(fs)
    h.set_option(MRC_OPT_EXACT_SPREAD, exact)
    try:
        for (a, b) in G.SHAPES:
            key = "%d_%d_%d" % (a, b, fs)
            blocks = np.array([G.pcm_to_float(r) for r in s["pcm_" + key]])
            smr, thr = h.smr(blocks, a, b, want_thresh=True)
            assert np.abs(thr - s["thr_" + key]).max() <= DB_ATOL, key
            assert np.abs(smr - s["smr_" + key]).max() <= DB_ATOL, key
    finally:
        h.set_option(MRC_OPT_EXACT_SPREAD, 0)


@pytest.mark.parametrize("tag,kind", [("single", "single"), ("nohuff", "nohuff"), ("jointch", "jointch"),
                                      ("joint", "joint"), ("indep", "indep"), ("jointlo", "joint"),
                                      ("jointtrain", "joint")])
def test_dropin_encode_chains_match_reference(tag, kind):
    import mrcaudiocodec_amd.codecThem as codec
    e = G.load("ref_encode.npz")
    assert list(e["table_order"]) == list(codec.TABLE_NAMES)
    G.check_chain(codec, e, tag, kind)


def test_decode_matches_reference(handles):
    e = G.load("ref_encode.npz")
    h = handles()
    for i in range(int(e["dec_n"])):
        k = "dec_%d" % i
        a, b = (int(v) for v in e[k + "_shape"])
        got = h.decode(a, b, e[k + "_os"][None], e[k + "_sf"][None], e[k + "_ba"][None],
                       np.stack([e[k + "_mant0"], e[k + "_mant1"]])[None], e[k + "_ms"][None])[0]
        want = e[k + "_jointdec"]
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), k
        one = h.decode(a, b, np.array([int(e[k + "_os1"])]), e[k + "_sf1"][None, None], e[k + "_ba1"][None, None],
                       e[k + "_mant1ch"][None, None])[0, 0]
        assert np.abs(one - e[k + "_dec"]).max() <= 1e-12 * np.abs(e[k + "_dec"]).max(), k


# ------------------------------------------------------------------------------------------------ file level
# tests/golden/ref_pac.npz: what the reference's command-line driver wrote for synthetic WAV files
# (tests/golden/make_golden_pac.py).  The product's CLI path -- WAV ingest, transient detector kernel, chained joint
# blocks on the device, Huffman + bit packing in C++ -- must write the same bytes, and its decoder the same samples.
@pytest.mark.parametrize("case", ["a48", "b44"])
def test_cli_pac_bytes_equal_reference_cli(tmp_path, case):
    from mrcaudiocodec_amd import cli
    g = G.load("ref_pac.npz")
    path = str(tmp_path / "in.wav")
    with open(path, "wb") as f:
        f.write(cli.wav_bytes(g[case + "_pcm"], int(g[case + "_rate"])))
    assert cli.encode_wav(path, use_huffman=True) == g[case + "_pac"].tobytes()
    assert cli.encode_wav(path, use_huffman=False) == g[case + "_pac_raw"].tobytes()
    assert cli.encode_wav(path, use_huffman=True, exact_spread=True) == g[case + "_pac"].tobytes()


@pytest.mark.parametrize("case", ["a48", "b44"])
def test_cli_decode_equals_reference_cli(tmp_path, case):
    from mrcaudiocodec_amd import cli
    g = G.load("ref_pac.npz")
    pac = str(tmp_path / "in.pac")
    with open(pac, "wb") as f:
        f.write(g[case + "_pac"].tobytes())
    pcm = cli.decode_pac_file(pac, str(tmp_path / "out.wav"))
    want = g[case + "_decoded"]             # its first 1024 samples are the reference driver's stale look-ahead block
    assert np.array_equal(pcm[:, :want.shape[1] - 1024], want[:, 1024:])
