"""
GPU parity tests of the decode side ("next" row f-4), through the C ABI: mrc_decode (codecThem.Decode / JointDecode),
mrc_dev_decode + mrc_dev_pcm16 behind pacfile.decode_pac, against oracle/decode.py.  Bars: windowed blocks and
overlap-added streams within 1e-12 of the block peak (the reference's IMDCT is an N-point inverse FFT, ours an
N/4-point one: same ~1e-13 as the forward transform), 16-bit PCM codes equal.
"""
import types

import numpy as np
import pytest

from oracle import decode as odec, fast, pacfile as opac

pytestmark = pytest.mark.gpu
SHAPES = [(1024, 1024), (128, 128), (1024, 128), (128, 1024)]


@pytest.fixture(scope="module")
def h():
    from mrcaudiocodec_amd import Handle
    hd = Handle()
    yield hd
    hd.close()


def _cp(a, b):
    cp = types.SimpleNamespace(a=a, b=b, nScaleBits=4, nMantSizeBits=4, nMDCTLines=1024, sampleRate=48000)
    cp.sfBands = fast.bands_for(a, b)
    return cp


def _random_block(rng, cp, nb, half):
    ba = rng.integers(0, 9, nb)
    ba[ba == 1] = 0
    ba[rng.integers(0, nb)] = 16                             # the widest mantissa
    sf = rng.integers(0, 16, nb)
    line_band = np.repeat(np.arange(nb), cp.sfBands.nLines)
    mant = (rng.integers(0, 1 << 16, half) & ((1 << np.maximum(ba[line_band], 1)) - 1)).astype(np.int32)
    mant[ba[line_band] == 0] = 0
    return sf.astype(np.int32), ba.astype(np.int32), mant


@pytest.mark.parametrize("shape", SHAPES)
def test_decode_mono_blocks(h, shape):
    a, b = shape
    cp = _cp(a, b)
    nb, half, n = cp.sfBands.nBands, (a + b) // 2, 12
    rng = np.random.default_rng(a * 3 + b)
    sf, ba, mant = map(np.stack, zip(*[_random_block(rng, cp, nb, half) for _ in range(n)]))
    osc = rng.integers(0, 16, n).astype(np.int32)
    got = h.decode(a, b, osc, sf[:, None, :], ba[:, None, :], mant[:, None, :])
    for i in range(n):
        want = odec.Decode(sf[i], ba[i], mant[i], int(osc[i]), cp)
        assert np.max(np.abs(got[i, 0] - want)) <= 1e-12 * max(np.max(np.abs(want)), 1e-300), (shape, i)


@pytest.mark.parametrize("shape", SHAPES)
def test_decode_joint_blocks(h, shape):
    a, b = shape
    cp = _cp(a, b)
    nb, half, n = cp.sfBands.nBands, (a + b) // 2, 8
    rng = np.random.default_rng(a * 5 + b)
    blocks = [[_random_block(rng, cp, nb, half) for _ in range(2)] for _ in range(n)]
    sf = np.array([[blk[c][0] for c in range(2)] for blk in blocks])
    ba = np.array([[blk[c][1] for c in range(2)] for blk in blocks])
    mant = np.array([[blk[c][2] for c in range(2)] for blk in blocks])
    osc = rng.integers(0, 16, (n, 4)).astype(np.int32)
    sw = rng.integers(0, 2, (n, nb)).astype(np.int32)
    got = h.decode(a, b, osc, sf, ba, mant, sw)
    for i in range(n):
        want = odec.JointDecode([sf[i, 0], sf[i, 1]], [ba[i, 0], ba[i, 1]], [mant[i, 0], mant[i, 1]], list(osc[i]), cp,
                                list(sw[i]))
        for c in range(2):
            assert np.max(np.abs(got[i, c] - want[c])) <= 1e-12 * max(np.max(np.abs(want[c])), 1e-300), (shape, i, c)


def test_dequantise_exact_through_flat_transform(h):
    # a single non-zero line makes every output sample one product dequantised * cos * window: checks the
    # dequantiser's value to the last bits through the transform (relative 1e-13)
    a = b = 128
    cp = _cp(a, b)
    nb, half = cp.sfBands.nBands, 128
    for (scale, bits, code) in [(0, 2, 1), (3, 5, 17), (15, 16, 40000), (7, 8, 255), (14, 3, 5), (15, 4, 8)]:
        sf = np.full(nb, scale, np.int32)
        ba = np.full(nb, bits, np.int32)
        mant = np.zeros(half, np.int32)
        mant[37] = code
        got = h.decode(a, b, [2], sf[None, None], ba[None, None], mant[None, None])[0, 0]
        want = odec.Decode(sf, ba, mant, 2, cp)
        assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want)) + 1e-300, (scale, bits, code)


@pytest.mark.parametrize("huff", [False, True])
def test_decode_pac_stream(h, huff):
    pytest.importorskip("torch")
    from mrcaudiocodec_amd import pacfile as ppac, synth
    x, shapes = synth.c4_transients(11)
    tone = synth.c1_sine(11)
    stream = np.stack([x + 0.3 * tone, 0.7 * x + 0.3 * tone])
    pac = opac.encode_stereo_stream(stream, shapes, huffman=huff)
    cp, want = odec.decode_pac(pac)
    nch, got = ppac.decode_pac(h, pac)
    got = got.cpu().numpy()
    assert nch == 2 and got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    pcm = ppac.decode_pac_pcm16(h, pac)
    assert np.array_equal(pcm, odec.pcm16(want[:, 1024:]))
    assert np.array_equal(h.pcm16(want), odec.pcm16(want))
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.5, -2.0, 0.5 / 65535, 1.5 / 65535, -1.5 / 65535, 0.999999, 1e-9])
    assert np.array_equal(h.pcm16(edge), odec.pcm16(edge))


def test_drop_in_decode_functions(h):
    from mrcaudiocodec_amd import codecThem as drop, synth
    from oracle import codec as ocodec
    cp = ocodec.default_params(nChannels=2)
    tone = synth.c1_sine(3)
    L, R = tone[:2048], 0.8 * tone[:2048]
    enc = ocodec.JointEncodeChannels(L, R, cp)
    sf, ba, mant, osc, sw = enc
    dense = [np.zeros(1024, np.int32), np.zeros(1024, np.int32)]
    for c in range(2):
        keep = np.repeat(np.asarray(ba[c]) > 0, cp.sfBands.nLines)
        dense[c][keep] = mant[c]
    want = odec.JointDecode(sf, ba, dense, osc, cp, sw)
    got = drop.JointDecode(sf, ba, dense, osc, cp, sw)
    for c in range(2):
        assert np.max(np.abs(got[c] - want[c])) <= 1e-12 * np.max(np.abs(want[c]))
    cp1 = ocodec.default_params(nChannels=1)
    sf1, ba1, m1, os1 = ocodec.EncodeSingleChannel(L, cp1)
    d1 = np.zeros(1024, np.int32)
    d1[np.repeat(np.asarray(ba1) > 0, cp1.sfBands.nLines)] = m1
    assert np.max(np.abs(drop.Decode(sf1, ba1, d1, os1, cp1) - odec.Decode(sf1, ba1, d1, os1, cp1))) <= 1e-12


def test_round_trip_on_device_at_scale(h):
    # size-independent property at a batch size the oracle could not decode in reasonable time: encode 512 stereo
    # frames of a tone pair on the GPU, pack, parse and decode on the GPU -> the signal comes back (SNR > 60 dB)
    pytest.importorskip("torch")
    from mrcaudiocodec_amd import pacfile as ppac, synth
    hops = 513
    tone = synth.c1_sine(hops)
    stream = np.stack([tone, 0.9 * tone])
    shapes = [(i * 1024, 1024, 1024) for i in range(hops - 1)]
    pac = ppac.encode_stereo_stream(h, stream, shapes, use_huffman=True)
    # (the chained call walks > 256 items of one stream here: its item ring refills; byte for byte the block-at-a-time loop)
    assert pac == ppac.encode_stereo_stream_per_block(h, stream, shapes, use_huffman=True)
    nch, x = ppac.decode_pac(h, pac)
    x = x.cpu().numpy()
    ref, dec = stream[:, 2048:(hops - 1) * 1024], x[:, 2048:(hops - 1) * 1024]
    snr = 10 * np.log10((ref ** 2).sum() / ((dec - ref) ** 2).sum())
    assert snr > 60, snr


def test_decode_mono_pac_file(h):
    # a one-channel file (header + WriteDataBlock chunks, pacfileThem.py:622-790): every block through the non-joint
    # reader; raw and Huffman-coded
    pytest.importorskip("torch")
    from mrcaudiocodec_amd import pacfile as ppac, synth
    hops = 9
    cfg = ppac.make_config()
    for huff in (False, True):
        # noise never lets a Huffman table win: the coded case is a pure tone
        x = synth.c1_sine(hops) + (0.0 if huff else 1.0) * synth.c2_noise(hops, seed=4, sigma=0.01)
        blocks = np.array(fast.blocks_from_stream(x, 1024))
        enc = h.encode_mono(blocks, 1024, 1024)
        data, _, table, _ = ppac.pack_blocks(cfg, 1024, 1024, enc["overall_scale"][:, None], enc["scale_factor"][:, None, :],
                                             enc["bit_alloc"][:, None, :], enc["mantissa"][:, None, :], huff)
        pac = ppac.header(cfg, 1, len(blocks) * 1024) + data.tobytes()
        cp, want = odec.decode_pac(pac)
        nch, got = ppac.decode_pac(h, pac)
        got = got.cpu().numpy()
        assert nch == 1 and cp.nChannels == 1 and got.shape == want.shape == (1, (len(blocks) + 1) * 1024)
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
        ref = x[1024:len(blocks) * 1024]
        snr = 10 * np.log10((ref ** 2).sum() / ((got[0, 1024:len(blocks) * 1024] - ref) ** 2).sum())
        assert snr > 15, snr
        if huff:
            assert (table != 15).any()
