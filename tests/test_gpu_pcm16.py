"""
GPU tests of the round-2 boundary additions, all through the C ABI:
  * 16-bit PCM ingest (pcmfile.py:91-100 applied on load) and the 16-bit mantissa plane -- same integers as the float64 /
    int32 path and as the oracle;
  * mrc_encode_stream_pcm16: host PCM -> host codes, chunks pipelined over three HIP streams;
  * mrc_encode_*_blocks: blocks of mixed shapes in one call;
  * the L1 names the drop-in gained (Bark, vQuantizeUniform, QuantizeUniform, Mantissa) and BitAlloc's in-place update of
    its SMR argument, against the golden vectors recorded from the reference's own modules.
"""
import os

import numpy as np
import pytest

import refgold as G
from oracle import fast

pytestmark = pytest.mark.gpu
INT_KEYS = ("overall_scale", "bit_alloc", "scale_factor", "mantissa", "reservoir_out")


@pytest.fixture(scope="module")
def h():
    from mrcaudiocodec_amd import Handle
    hd = Handle(device_id=0)
    yield hd
    hd.close()


def _pcm(n, seed, sigma=0.1):
    return np.clip(np.rint(np.random.default_rng(seed).normal(0, sigma * 32767, n)), -32768, 32767).astype(np.int16)


def test_pcm_to_float_every_code(h):
    codes = np.arange(-32768, 32768, dtype=np.int16)
    want = np.where(codes == -32768, 0.0, 2.0 * codes.astype(np.float64) / 65535.0)      # one IEEE division (quantize.py:108)
    got = h.pcm_to_float(codes)
    assert np.array_equal(got, want)
    assert not np.signbit(got[0])                                                          # -32768 -> +0.0
    e = G.load("ref_encode.npz")
    assert np.array_equal(h.pcm_to_float(e["pcmmap_in"]), e["pcmmap_out"])                 # the reference's own values


@pytest.mark.parametrize("joint", [False, True])
def test_int16_ingest_and_mantissa16_equal_float_path(h, joint):
    torch = pytest.importorskip("torch")
    from mrcaudiocodec_amd.batch import StreamEncoder
    enc = StreamEncoder(h)
    F = 200
    pl = _pcm((F + 1) * 1024, 11)
    pl[5000:5010] = [-32768, 32767, -32767, 0, 1, -1, 2, -2, -32768, 12345]
    pr = (0.8 * pl + 0.2 * _pcm((F + 1) * 1024, 12)).astype(np.int16) if joint else None
    dev = enc.device
    tl16 = torch.from_numpy(pl).to(dev)
    tl64 = torch.from_numpy(G.pcm_to_float(pl)).to(dev)
    tr16 = torch.from_numpy(pr).to(dev) if joint else None
    tr64 = torch.from_numpy(G.pcm_to_float(pr)).to(dev) if joint else None
    ref = {k: v.cpu().numpy() for k, v in enc.encode_long(tl64, tr64, F).items()}
    got = {k: v.cpu().numpy() for k, v in enc.encode(1024, 1024, tl16, tr16, F, 1024, mantissa16=True, fresh=True).items()}
    got["mantissa"] = got["mantissa"].view(np.uint16).astype(np.int32)
    for k in INT_KEYS + (("ms_switch",) if joint else ()):
        assert np.array_equal(got[k], ref[k]), k
    # and against the oracle on the converted samples
    bl = np.array(fast.blocks_from_stream(G.pcm_to_float(pl), 1024))[:48]
    if joint:
        br = np.array(fast.blocks_from_stream(G.pcm_to_float(pr), 1024))[:48]
        want = fast.encode_joint_batch(bl, br, 1024, 1024)
    else:
        want = fast.encode_mono_batch(bl, 1024, 1024)
    for k in INT_KEYS + (("ms_switch",) if joint else ()):
        assert np.array_equal(np.squeeze(got[k][:48]), np.squeeze(want[k])), k
    # explicit offsets (even and odd sample offsets; short and transition shapes) from int16
    for (a, b) in ((1024, 1024), (1024, 128), (128, 128), (128, 1024)):
        offs = np.array([0, 1, 2048, 4097, 9000, 12345], dtype=np.int64)
        to = torch.from_numpy(offs).to(dev)
        r64 = {k: v.cpu().numpy() for k, v in enc.encode(a, b, tl64, tr64, len(offs), 0, to, fresh=True).items()}
        r16 = {k: v.cpu().numpy() for k, v in enc.encode(a, b, tl16, tr16, len(offs), 0, to, fresh=True).items()}
        for k in INT_KEYS + (("ms_switch",) if joint else ()):
            assert np.array_equal(r16[k], r64[k]), (a, b, k)


@pytest.mark.parametrize("joint", [False, True])
@pytest.mark.parametrize("pinned", [False, True])
def test_encode_stream_pcm16_pipeline(h, joint, pinned):
    from mrcaudiocodec_amd import PinnedArray
    F = 150
    pl = _pcm((F + 1) * 1024, 21)
    pr = (0.8 * pl + 0.2 * _pcm((F + 1) * 1024, 22)).astype(np.int16) if joint else None
    pl[:1024] = 0
    if joint:
        pr[:1024] = 0
    res_in = np.random.default_rng(3).integers(-50, 300, F).astype(np.int32)
    keep = []
    out = None
    if pinned:
        def pin(a):
            p = PinnedArray(a.shape, a.dtype); p.array[...] = a; keep.append(p); return p.array
        pl = pin(pl)
        pr = pin(pr) if joint else None
        nb = len(h.bands(1024, 1024))
        ns, nsig = (2, 4) if joint else (1, 1)
        shapes = dict(overall_scale=((F, nsig), np.int32), scale_factor=((F, ns, nb), np.int32),
                      bit_alloc=((F, ns, nb), np.int32), mantissa=((F, ns, 1024), np.uint16), reservoir_out=((F,), np.int32))
        if joint:
            shapes["ms_switch"] = ((F, nb), np.int32)
        out = {}
        for k, (shape, dt) in shapes.items():
            p = PinnedArray(shape, dt); keep.append(p); out[k] = p.array
    got = h.encode_stream_pcm16(pl, pr, res_in, chunk_frames=17, out=out)          # 9 chunks over the ring of 4 chunk buffers
    bl = np.array(fast.blocks_from_stream(G.pcm_to_float(pl), 1024))
    if joint:
        br = np.array(fast.blocks_from_stream(G.pcm_to_float(pr), 1024))
        want = fast.encode_joint_batch(bl, br, 1024, 1024, reservoir_in=res_in)
    else:
        want = fast.encode_mono_batch(bl, 1024, 1024, reservoir_in=res_in)
    for k in INT_KEYS + (("ms_switch",) if joint else ()):
        assert np.array_equal(np.squeeze(got[k]).astype(np.int64), np.squeeze(want[k]).astype(np.int64)), k
    one = h.encode_stream_pcm16(pl, pr, res_in, chunk_frames=0)                     # a single chunk: same results
    for k in got:
        assert np.array_equal(one[k], got[k]), k
    from mrcaudiocodec_amd import MrcError
    with pytest.raises(ValueError):
        h.encode_stream_pcm16(pl, pr, res_in[:5])                                   # reservoir_in of the wrong length
    with pytest.raises(ValueError):
        h.encode_mono(bl[:4], 1024, 1024, [0])
    for p in keep:
        p.free()


@pytest.mark.parametrize("joint", [False, True])
def test_encode_blocks_of_mixed_shapes(h, joint):
    from mrcaudiocodec_amd import synth
    x, shapes = synth.c4_transients(12)
    stream = np.stack([x, 0.7 * x + 0.01 * synth.c2_noise(12, seed=3)[:len(x)]])
    a = [s[1] for s in shapes]; b = [s[2] for s in shapes]
    left = [stream[0, o:o + aa + bb] for (o, aa, bb) in shapes]
    right = [stream[1, o:o + aa + bb] for (o, aa, bb) in shapes] if joint else None
    res_in = np.arange(len(shapes), dtype=np.int32) % 40
    got = h.encode_blocks(left, a, b, right, res_in)
    assert len(set(zip(a, b))) == 4
    for i, (o, aa, bb) in enumerate(shapes):
        nb, half = len(h.bands(aa, bb)), (aa + bb) // 2
        if joint:
            one = h.encode_joint(left[i][None], right[i][None], aa, bb, [int(res_in[i])])
            assert np.array_equal(got["ms_switch"][i, :nb], one["ms_switch"][0]) and not got["ms_switch"][i, nb:].any()
            assert np.array_equal(got["overall_scale"][i], one["overall_scale"][0])
            sf, ba, m = one["scale_factor"][0], one["bit_alloc"][0], one["mantissa"][0]
        else:
            one = h.encode_mono(left[i][None], aa, bb, [int(res_in[i])])
            assert got["overall_scale"][i] == one["overall_scale"][0]
            sf, ba, m = one["scale_factor"], one["bit_alloc"], one["mantissa"]
        assert got["reservoir_out"][i] == one["reservoir_out"][0]
        assert np.array_equal(got["scale_factor"][i, :, :nb], sf) and not got["scale_factor"][i, :, nb:].any()
        assert np.array_equal(got["bit_alloc"][i, :, :nb], ba) and not got["bit_alloc"][i, :, nb:].any()
        assert np.array_equal(got["mantissa"][i, :, :half], m) and not got["mantissa"][i, :, half:].any()
    with pytest.raises(ValueError):
        h.encode_blocks(left, a[:-1], b, right)


def test_dropin_l1_names_added_in_round_2(golden_dir):
    import mrcaudiocodec_amd.codecThem as codec
    q = np.load(os.path.join(golden_dir, "quantize.npz"), allow_pickle=False)
    vals = q["vals"]
    for nb, want, want_scalar in zip(q["nbits_cases"], q["vquant_out"], q["quant_out"]):
        assert np.array_equal(codec.vQuantizeUniform(vals, int(nb)), want), nb
        assert [codec.QuantizeUniform(float(v), int(nb)) for v in vals[:40]] == [int(w) for w in want_scalar[:40]], nb
    for (sc, sb, mb), want in zip(q["mant_cases"][::7], q["mant_out"][::7]):
        assert [codec.Mantissa(float(v), int(sc), int(sb), int(mb)) for v in vals[:60]] == [int(w) for w in want[:60]]
    p = G.load("ref_psychoac.npz")
    assert np.abs(codec.Bark(p["f_in"]) - p["bark_out"]).max() <= 1e-13
    assert abs(codec.Bark(1000.0) - float(p["bark_out"][list(p["f_in"]).index(1000.0)])) <= 1e-13
    b = np.load(os.path.join(golden_dir, "bitalloc.npz"), allow_pickle=False)
    for i in range(int(b["n"])):
        smr = b["smr_%d" % i].copy()
        bits, left = codec.BitAlloc(float(b["budget_%d" % i]), int(b["maxb_%d" % i]), len(smr), b["nlines_%d" % i], smr)
        assert np.array_equal(bits, b["bits_%d" % i]) and left == int(b["left_%d" % i]), i
        assert np.array_equal(smr, b["smr_after_%d" % i]), i          # bitalloc.py:132-151: updated in place


@pytest.mark.parametrize("exact", [0, 1])
def test_exact_ties_between_streams(h, exact):
    """R == L (S == 0: the side stream is digital silence, every SMR sits on the same floor value) and R == -L: the
    joint bit allocation then runs on exact ties -- np.argmax's first-maximum rule decides -- in both spreading modes."""
    from mrcaudiocodec_amd import synth
    s = synth.c3_stereo(24)
    bl = np.array(fast.blocks_from_stream(s[0], 1024))
    h.set_option(1, exact)
    try:
        for br in (bl, -bl, np.zeros_like(bl)):
            got = h.encode_joint(bl, br, 1024, 1024)
            want = fast.encode_joint_batch(bl, br, 1024, 1024)
            for k in INT_KEYS + ("ms_switch",):
                assert np.array_equal(got[k], want[k]), k
    finally:
        h.set_option(1, 0)


def test_two_shards_give_the_bytes_of_the_unsharded_stream(h):
    """SURVEY.md 8(e) / configs[4]: a stereo stream cut into contiguous frame ranges (each with its one-hop halo, each
    generated and encoded on its own as a rank would) yields, block for block, the `.pac` chunks of the unsharded
    encode -- independent-frames mode, joint path, Huffman + bit packing on the host."""
    torch = pytest.importorskip("torch")
    import bench
    from mrcaudiocodec_amd import pacfile as ppac
    from mrcaudiocodec_amd.batch import StreamEncoder
    from mrcaudiocodec_amd.shard import shard_frames
    enc = StreamEncoder(h)
    cfg = ppac.make_config()
    F = 96

    def chunks(first, n):
        sl, sr = bench.stream_slice(torch, enc.device, "c3", first, n)
        out = {k: v.cpu().numpy() for k, v in enc.encode_long(sl, sr, n, mantissa16=True).items()}
        mant = out["mantissa"].view(np.uint16).astype(np.int32)
        data, offs, table, saved = ppac.pack_joint_blocks(cfg, 1024, 1024, out["overall_scale"], out["ms_switch"],
                                                          out["scale_factor"], out["bit_alloc"], mant, use_huffman=True)
        return data.tobytes(), out
    whole, out = chunks(0, F)
    assert 0 < out["ms_switch"].mean() < 1
    for world in (2, 3):
        parts = [chunks(*shard_frames(F, world, r))[0] for r in range(world)]
        assert b"".join(parts) == whole, world


def test_joint_smr_skip_changes_no_output(h):
    """smr_kernel leaves out the (signal, band) pairs the M/S switch does not select (ms_stereo.py:70-81: they never
    reach the bit allocation).  With MRC_OPT_SMR_ALL_BANDS every SMR is computed: every output must be the same, on
    content with mixed decisions, all-M/S and all-L/R bands, for every block shape."""
    from mrcaudiocodec_amd import synth
    MRC_OPT_SMR_ALL_BANDS = 2
    s = synth.c3_stereo(40)
    cases = [(s[0], s[1]), (s[0], 0.9 * s[0]), (s[0], synth.c2_noise(40, seed=99) * 3.0)]
    for (a, b) in ((1024, 1024), (1024, 128), (128, 128), (128, 1024)):
        for L, R in cases:
            offs = np.arange(0, 30) * 1024 + 512
            bl = np.stack([L[o:o + a + b] for o in offs]); br = np.stack([R[o:o + a + b] for o in offs])
            got = h.encode_joint(bl, br, a, b)
            h.set_option(MRC_OPT_SMR_ALL_BANDS, 1)
            try:
                full = h.encode_joint(bl, br, a, b)
            finally:
                h.set_option(MRC_OPT_SMR_ALL_BANDS, 0)
            for k in INT_KEYS + ("ms_switch",):
                assert np.array_equal(got[k], full[k]), (a, b, k)
    assert 0 < got["ms_switch"].mean() <= 1
