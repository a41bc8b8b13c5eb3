"""
GPU parity of the chained stream encode (mrc_encode_chained_stream_pac: phase A batched over all blocks, the
reservoir-dependent back end as a serial scan per stream on the device, device packer) -- through the C ABI, against
the oracle's restatement of the reference's encode loop (pacfileThem.py:1159-1214, 973-984; codecThem.py:262-278,
381-396, 503) and against this package's own block-at-a-time form of the same loop.  Byte for byte.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def h():
    from mrcaudiocodec_amd import Handle
    hd = Handle(device_id=0)
    yield hd
    hd.close()


def _switching_stream(hops=16, seed=42):
    from mrcaudiocodec_amd import synth
    x, shapes = synth.c4_transients(hops, seed=seed)
    g = synth.c2_noise(hops, seed=seed + 1, sigma=0.05)
    tone = synth.c1_sine(hops)
    return np.stack([x + 0.3 * tone, 0.7 * x + 0.3 * tone + 0.05 * g]), shapes


@pytest.mark.parametrize("huff", [True, False])
def test_chained_bytes_equal_oracle_and_per_block_loop(h, huff):
    from mrcaudiocodec_amd import pacfile as ppac
    from oracle import pacfile as opac
    stream, shapes = _switching_stream()
    assert shapes[-1][2] == 1024 and len({(a, b) for (_, a, b) in shapes}) == 4      # all four block shapes occur
    got = ppac.encode_stereo_stream(h, stream, shapes, use_huffman=huff)
    assert got == opac.encode_stereo_stream(stream, shapes, huffman=huff)
    assert got == ppac.encode_stereo_stream_per_block(h, stream, shapes, use_huffman=huff)


def test_reservoir_trace_follows_the_reference_chain(h):
    # the reservoir after every block = what codecThem.JointEncode leaves in codingParams.bitReservoir (lines 274, 503),
    # then Encode's two channels of Close() (224, 332)
    from oracle import codec as ocodec
    stream, shapes = _switching_stream(hops=11, seed=7)
    r = h.encode_chained_pac(stream[0][None], stream[1][None], [shapes], want_trace=True)
    cp = ocodec.default_params(nChannels=2)
    cp.bitReservoir = 0
    want = []
    for (off, a, b) in shapes:
        cp.a, cp.b = a, b
        cp.sfBands = ocodec.bands_for_block(a, b, cp.nMDCTLines, cp.sampleRate)
        ocodec.JointEncode([stream[0][off:off + a + b].copy(), stream[1][off:off + a + b].copy()], cp)
        want.append(cp.bitReservoir)
    off, a, b = shapes[-1]
    cp.a = cp.b = 1024
    cp.sfBands = ocodec.bands_for_block(1024, 1024, cp.nMDCTLines, cp.sampleRate)
    for ch in range(2):
        one = ocodec.default_params(nChannels=1)
        one.bitReservoir = cp.bitReservoir
        ocodec.Encode([np.concatenate([stream[ch][off + a:off + a + b], np.zeros(1024)])], one)
        cp.bitReservoir = one.bitReservoir
        want.append(cp.bitReservoir)
    assert r["reservoir_trace"].tolist() == want
    assert int(r["reservoir_out"][0]) == want[-1]


def test_event_order_repair_pass_gives_the_same_bytes(h):
    # chain_prep_kernel derives the order of the bit allocation's grant attempts from the 6 dB periodicity of the running
    # SMRs and CHECKS it against the keys; the repair pass behind that check only runs on near-ties in production.  Here
    # the candidate order is scrambled on purpose so that the repair has to restore it.
    from mrcaudiocodec_amd import pacfile as ppac
    stream, shapes = _switching_stream(hops=11, seed=5)
    want = ppac.encode_stereo_stream(h, stream, shapes)
    h.set_option(3, 1)
    try:
        got = ppac.encode_stereo_stream(h, stream, shapes)
    finally:
        h.set_option(3, 0)
    assert got == want


def test_many_streams_with_different_schedules(h):
    from mrcaudiocodec_amd import pacfile as ppac, synth
    from oracle import pacfile as opac
    hops = 11
    x, sh_sw = synth.c4_transients(hops)
    tone = synth.c1_sine(hops)
    g = synth.c2_noise(hops, seed=3, sigma=0.05)
    sh_long = [(i * 1024, 1024, 1024) for i in range(hops - 1)]
    n = len(tone)
    streams = np.stack([
        np.stack([x + 0.3 * tone, 0.7 * x + 0.3 * tone + 0.05 * g])[:, :n],
        np.stack([tone, 0.9 * tone]),
        np.stack([0.5 * tone + g, 0.5 * tone - g]),
        np.zeros((2, n)),                                              # digital silence: the reservoir only grows
        np.stack([g, 0.2 * tone]),
    ])
    streams = np.concatenate([streams, np.stack([0.4 * tone, g])[None]])     # ... and a stream of ONE block (+ Close())
    shapes = [sh_sw, sh_long, sh_long[:6], sh_long[:4], sh_sw, sh_long[:1]]
    got = ppac.encode_stereo_streams(h, streams, shapes)
    for s in range(len(shapes)):
        assert got[s] == opac.encode_stereo_stream(streams[s], shapes[s], huffman=True), s


_LONG_SWITCHED = {}


def _long_switched():
    """a block-switched stereo stream of 601 hops (all four block shapes, > 256 items: the scan's item ring refills while the
    shapes change) and its bytes from the block-at-a-time loop; built once for the parametrised test below"""
    if not _LONG_SWITCHED:
        stream, shapes = _switching_stream(hops=601, seed=9)
        _LONG_SWITCHED.update(stream=stream, shapes=shapes)
    return _LONG_SWITCHED


@pytest.mark.parametrize("threads", [256, 512, 1024])
def test_scan_workgroup_sizes_give_the_per_block_bytes(h, threads):
    # chain_phase_b_kernel exists for 256 / 512 / 1024 threads per stream (two / one / one units of four lines per thread,
    # different event staging and Huffman accumulators); the library picks 512 for <= 512 streams and 256 above, so the
    # small-batch tests never run the 256-thread form.  Every form, forced through MRC_OPT_CHAIN_THREADS, on a long
    # block-switched stream: the bytes and the reservoir after every block equal the block-at-a-time loop's / the oracle's.
    from mrcaudiocodec_amd import pacfile as ppac
    from oracle import pacfile as opac
    d = _long_switched()
    stream, shapes = d["stream"], d["shapes"]
    assert len(shapes) > 700 and len({(a, b) for (_, a, b) in shapes}) == 4
    if "want" not in d:
        d["want"] = ppac.encode_stereo_stream_per_block(h, stream, shapes, use_huffman=True)
        n40 = 40
        d["prefix"] = opac.encode_stereo_stream(stream[:, :shapes[n40 - 1][0] + shapes[n40 - 1][1] + shapes[n40 - 1][2]],
                                                shapes[:n40], huffman=True)
    h.set_option(4, threads)
    try:
        got = ppac.encode_stereo_stream(h, stream, shapes, use_huffman=True)
        r = h.encode_chained_pac(stream[0][None], stream[1][None], [shapes], want_trace=True)
        short = ppac.encode_stereo_stream(h, stream, shapes[:40], use_huffman=True)
    finally:
        h.set_option(4, 0)
    assert got == d["want"]
    assert short == d["prefix"]                                       # ... and the oracle on a prefix
    trace = np.asarray(r["reservoir_trace"])
    if "trace" not in d:
        d["trace"] = trace
    assert np.array_equal(trace, d["trace"])                          # the same reservoir after every block in every form


def test_many_streams_take_the_256_thread_scan(h):
    # > 512 streams: the library's own choice is the 256-thread scan.  600 streams made of 6 different ones (different
    # block-shape schedules and lengths), each file byte for byte what the stream gives when encoded alone.
    from mrcaudiocodec_amd import pacfile as ppac, synth
    hops = 9
    x, sh_sw = synth.c4_transients(hops)
    tone = synth.c1_sine(hops)
    g = synth.c2_noise(hops, seed=3, sigma=0.05)
    sh_long = [(i * 1024, 1024, 1024) for i in range(hops - 1)]
    n = len(tone)
    base = [np.stack([x + 0.3 * tone, 0.7 * x + 0.3 * tone + 0.05 * g])[:, :n], np.stack([tone, 0.9 * tone]),
            np.stack([0.5 * tone + g, 0.5 * tone - g]), np.zeros((2, n)), np.stack([g, 0.2 * tone]),
            np.stack([0.4 * tone, g])]
    base_shapes = [sh_sw, sh_long, sh_long[:6], sh_long[:4], sh_sw, sh_long[:1]]
    alone = [ppac.encode_stereo_stream(h, base[i], base_shapes[i]) for i in range(6)]
    pick = (np.arange(600) * 7) % 6
    got = ppac.encode_stereo_streams(h, np.stack([base[i] for i in pick]), [base_shapes[i] for i in pick])
    assert h.get_option(4) == 0
    for s, i in enumerate(pick):
        assert got[s] == alone[i], (s, i)


@pytest.mark.parametrize("slab", [7, 64, 100000])
def test_slabs_do_not_change_the_bytes(h, slab):
    # MRC_OPT_CHAIN_SLAB_BLOCKS: a call is cut into slabs of at most `slab` blocks (whole streams while they fit, a longer
    # stream alone in time slabs, the reservoir carried across) so that its device memory is bounded by the slab; bytes, stream
    # and item offsets, reservoirs and the reservoir trace are those of the unslabbed call -- through the host-memory entry
    # and the device-memory one
    torch = pytest.importorskip("torch")
    from mrcaudiocodec_amd import synth
    from mrcaudiocodec_amd.batch import StreamEncoder
    d = _long_switched()
    hops = 11
    x, sh_sw = synth.c4_transients(hops)
    tone = synth.c1_sine(hops)
    g = synth.c2_noise(hops, seed=3, sigma=0.05)
    sh_long = [(i * 1024, 1024, 1024) for i in range(hops - 1)]
    n = len(tone)
    few = np.stack([np.stack([x + 0.3 * tone, 0.7 * x + 0.3 * tone + 0.05 * g])[:, :n], np.stack([tone, 0.9 * tone]),
                    np.stack([0.5 * tone + g, 0.5 * tone - g]), np.stack([g, 0.2 * tone])])
    few_shapes = [sh_sw, sh_long, sh_long[:6], sh_long[:3]]
    cases = [(d["stream"][:, None, :], [d["shapes"]]), (np.ascontiguousarray(few.transpose(1, 0, 2)), few_shapes)]
    for (st, shapes) in cases:
        kw = dict(want_trace=True, want_items=True, num_samples=[sum(b for (_, _, b) in sh) for sh in shapes], reservoir_in=[5] * len(shapes))
        h.set_option(6, 0)
        want = h.encode_chained_pac(st[0], st[1], shapes, **kw)
        h.set_option(6, slab)
        try:
            got = h.encode_chained_pac(st[0], st[1], shapes, **kw)
            enc = StreamEncoder(handle=h)
            dl, dr = (torch.from_numpy(np.ascontiguousarray(st[c])).cuda() for c in (0, 1))
            dev = enc.encode_chained_pac(dl, dr, shapes, num_samples=kw["num_samples"], reservoir_in=kw["reservoir_in"], want_items=True)
        finally:
            h.set_option(6, 131072)
        assert got["bytes"].tobytes() == want["bytes"].tobytes()
        for k in ("stream_offset", "item_offset", "reservoir_out", "reservoir_trace"):
            assert np.array_equal(got[k], want[k]), k
        assert dev["bytes"].cpu().numpy().tobytes() == want["bytes"].tobytes()
        assert np.array_equal(dev["stream_offset"], want["stream_offset"]) and np.array_equal(dev["item_offset"], want["item_offset"])
        assert np.array_equal(dev["reservoir_out"], want["reservoir_out"])


def test_output_beyond_the_first_buffer_is_fetched(h):
    # the binding's first buffer holds ~1 KB per block; at 12 bits per sample a block packs to more: the call reports the size
    # and the bytes -- complete in the handle's device buffer -- are fetched (mrc_chain_fetch_output), not encoded again
    from mrcaudiocodec_amd import Handle, pacfile as ppac
    hd = Handle(target_bits_per_sample=12.0)
    try:
        stream, shapes = _switching_stream(hops=12, seed=3)
        got = ppac.encode_stereo_stream(hd, stream, shapes)
        assert len(got) > len(shapes) * 1024 + 2 * 4096
        assert got == ppac.encode_stereo_stream_per_block(hd, stream, shapes)
        hd.set_option(6, 5)                                            # several slabs: nothing to fetch, the binding encodes again
        assert ppac.encode_stereo_stream(hd, stream, shapes) == got
    finally:
        hd.close()


def test_pcm16_input_equals_float_input(h):
    from mrcaudiocodec_amd import synth
    rng = np.random.default_rng(11)
    hops = 9
    pcm = np.clip(np.rint(rng.normal(0, 0.08 * 32767, (2, (hops + 1) * 1024))), -32768, 32767).astype(np.int16)
    pcm[:, :1024] = 0
    pcm[0, 5000] = -32768                                              # the code the reader maps to 0.0
    shapes = [(i * 1024, 1024, 1024) for i in range(hops)]
    a = h.encode_chained_pac(pcm[0][None], pcm[1][None], [shapes], num_samples=[hops * 1024])
    x = synth.pcm_to_float(pcm)
    b = h.encode_chained_pac(x[0][None], x[1][None], [shapes], num_samples=[hops * 1024])
    assert a["bytes"].tobytes() == b["bytes"].tobytes()
    assert np.array_equal(a["stream_offset"], b["stream_offset"])


@pytest.mark.parametrize("tbps,res_in", [(0.02, 0), (2.86, 200000), (2.86, -3000), (0.5, 17)])
def test_budget_extremes(tbps, res_in):
    # budgets at or below zero (nothing is granted, the negative remainder is carried), and a reservoir so large that
    # every band runs into the 16-bit cap (the whole event list is consumed)
    from mrcaudiocodec_amd import Handle, pacfile as ppac
    from oracle import codec as ocodec
    hd = Handle(device_id=0, target_bits_per_sample=tbps)
    try:
        stream, shapes = _switching_stream(hops=11, seed=3)
        r = hd.encode_chained_pac(stream[0][None], stream[1][None], [shapes], reservoir_in=[res_in], want_trace=True,
                                  with_flush=False)
        cp = ocodec.default_params(nChannels=2, targetBitsPerSample=tbps)
        cp.bitReservoir = res_in
        want = []
        for (off, a, b) in shapes:
            cp.a, cp.b = a, b
            cp.sfBands = ocodec.bands_for_block(a, b, cp.nMDCTLines, cp.sampleRate)
            ocodec.JointEncode([stream[0][off:off + a + b].copy(), stream[1][off:off + a + b].copy()], cp)
            want.append(cp.bitReservoir)
        assert r["reservoir_trace"].tolist() == want
    finally:
        hd.close()


def test_argument_checks(h):
    from mrcaudiocodec_amd._lib import MrcError
    stream, shapes = _switching_stream(hops=6, seed=1)
    with pytest.raises(MrcError):                                       # a block that reaches outside the stream
        h.encode_chained_pac(stream[0][None], stream[1][None], [[(len(stream[0]) - 1024, 1024, 1024)]])
    with pytest.raises(MrcError):                                       # not one of the four shapes
        h.encode_chained_pac(stream[0][None], stream[1][None], [[(0, 512, 512)]])
    with pytest.raises(MrcError):                                       # Close() needs a long last block
        h.encode_chained_pac(stream[0][None], stream[1][None], [[(0, 1024, 128)]], with_flush=True)



def test_item_offsets_are_optional_and_consistent(h):
    # want_items: where every block's chunk pair starts (read back from the packer's positions); without it only the
    # stream starts come back (gathered on the device)
    stream, shapes = _switching_stream(hops=9, seed=2)
    two = np.stack([stream, 0.5 * stream])
    ns = [shapes[-1][0] + shapes[-1][1]] * 2
    a = h.encode_chained_pac(two[:, 0], two[:, 1], [shapes, shapes], num_samples=ns, want_items=True)
    b = h.encode_chained_pac(two[:, 0], two[:, 1], [shapes, shapes], num_samples=ns)
    assert b["item_offset"] is None and a["bytes"].tobytes() == b["bytes"].tobytes()
    assert np.array_equal(a["stream_offset"], b["stream_offset"])
    n_items = len(shapes) + 2
    io = a["item_offset"]
    assert len(io) == 2 * n_items + 1 and np.all(np.diff(io) > 0) and io[-1] == a["total"]
    hdr = int(io[0] - a["stream_offset"][0])
    assert hdr > 0 and io[n_items] - hdr == a["stream_offset"][1]      # the header sits between the stream start and item 0
