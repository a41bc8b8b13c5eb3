"""
GPU tests of the device-side `.pac` chunk packer (mrc_dev_pack_blocks, csrc/mrc_kernels_pack.hip): byte for byte the
output of the host packer (csrc/mrc_pack.cpp), which tests/test_reference_golden.py pins to the bytes the
reference's own command-line driver wrote -- and, for the long joint blocks of that fixture, the reference's bytes
directly.  Everything goes through the C ABI.
"""
import numpy as np
import pytest

import refgold as G

pytestmark = pytest.mark.gpu
SHAPES = [(1024, 1024), (1024, 128), (128, 128), (128, 1024)]


@pytest.fixture(scope="module")
def env():
    import torch
    from mrcaudiocodec_amd import pacfile
    from mrcaudiocodec_amd.batch import StreamEncoder
    enc = StreamEncoder(device_id=0)
    c = enc.h.cfg
    cfg = pacfile.make_config(c.sample_rate, c.n_mdct_lines, c.n_short, c.n_scale_bits, c.n_mant_size_bits,
                              c.target_bits_per_sample, c.blksw_bits_a, c.blksw_bits_b)
    return torch, pacfile, enc, cfg


def _dev(torch, arr, dtype):
    return torch.as_tensor(np.ascontiguousarray(arr), device="cuda:0").to(dtype).contiguous()


def _host_pack(pacfile, cfg, a, b, joint, o, use_huffman, huff_table=None):
    if joint:
        return pacfile.pack_joint_blocks(cfg, a, b, o["overall_scale"], o["ms_switch"], o["scale_factor"], o["bit_alloc"],
                                         o["mantissa"], use_huffman, huff_table)
    return pacfile.pack_blocks(cfg, a, b, o["overall_scale"], o["scale_factor"], o["bit_alloc"], o["mantissa"], use_huffman,
                               huff_table)


def _check(torch, pacfile, enc, cfg, a, b, joint, dev_out, use_huffman, given=None):
    host = {k: v.cpu().numpy() for k, v in dev_out.items() if v is not None}
    if host["mantissa"].dtype == np.int16:
        host["mantissa"] = host["mantissa"].view(np.uint16)
    if not joint and host["overall_scale"].ndim == 1:
        host["overall_scale"] = host["overall_scale"][:, None]
    want_bytes, want_offs, want_table, want_saved = _host_pack(pacfile, cfg, a, b, joint, host, use_huffman,
                                                               None if given is None else given.cpu().numpy())
    got = enc.pack(a, b, dev_out, use_huffman=use_huffman, huff_table=given)
    assert np.array_equal(got["block_offset"].cpu().numpy(), want_offs)
    assert np.array_equal(got["huff_table"].cpu().numpy(), np.asarray(want_table).reshape(got["huff_table"].shape))
    if given is None:
        assert np.array_equal(got["bits_saved"].cpu().numpy(), want_saved)
    gb = got["bytes"].cpu().numpy()
    assert gb.shape == want_bytes.shape
    assert np.array_equal(gb, want_bytes), "first differing byte %d" % int(np.nonzero(gb != want_bytes)[0][0])
    return got


@pytest.mark.parametrize("joint", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_device_pack_equals_host_pack_on_encoder_output(env, shape, joint):
    """blocks of every shape, encoded on the device (noise of varying level: bands with and without bits, all four
    Huffman pricing mostly ends at raw chunks here; the tables are exercised below), both mantissa formats, priced here / raw / tables given"""
    torch, pacfile, enc, cfg = env
    a, b = shape
    rng = np.random.default_rng(a * 7 + b + joint)
    n = 96
    level = 10.0 ** rng.uniform(-4, -0.3, n)
    left = (rng.normal(0, 1, (n, a + b)) * level[:, None]).clip(-1, 1)
    right = (0.7 * left + 0.3 * rng.normal(0, 1, (n, a + b)) * level[:, None]).clip(-1, 1) if joint else None
    offs = torch.arange(n, dtype=torch.int64, device="cuda:0") * (a + b)
    L = _dev(torch, left.reshape(-1), torch.float64)
    R = _dev(torch, right.reshape(-1), torch.float64) if joint else None
    res = _dev(torch, rng.integers(0, 200, n), torch.int32)
    for m16 in (False, True):
        out = dict(enc.encode(a, b, L, R, n, 0, offs, res, mantissa16=m16))
        got = _check(torch, pacfile, enc, cfg, a, b, joint, out, True)
        _check(torch, pacfile, enc, cfg, a, b, joint, out, False)
        table, _, _ = enc.huffman_gain(a, b, out) if not m16 else (got["huff_table"], None, None)
        _check(torch, pacfile, enc, cfg, a, b, joint, out, True, given=table.contiguous())


@pytest.mark.parametrize("joint", [False, True])
def test_device_pack_on_crafted_codes(env, joint):
    """codes the encoder seldom produces: 16-bit mantissas, values beyond every table (escape + raw), the escape values
    themselves, empty and full bit allocations, every scale factor"""
    torch, pacfile, enc, cfg = env
    a = b = 1024
    nb = len(enc.h.bands(a, b))
    nch = 2 if joint else 1
    rng = np.random.default_rng(5 + joint)
    n = 64
    ba = rng.integers(0, 17, (n, nch, nb)).astype(np.int32)
    ba[ba == 1] = 0                                                  # the allocator never leaves a single bit (bitalloc.py:141-150)
    ba[0] = 0
    ba[1] = 16
    sf = rng.integers(0, 16, (n, nch, nb)).astype(np.int32)
    lines = enc.h.bands(a, b)
    band_of = np.repeat(np.arange(nb), lines)
    bits = ba[:, :, band_of]                                         # [n][nch][1024]
    mant = (rng.integers(0, 1 << 16, bits.shape) & ((1 << bits) - 1)).astype(np.int32)
    small = rng.random(bits.shape) < 0.6                             # most codes small: inside the tables
    mant = np.where(small, mant % 20, mant) & ((1 << bits) - 1)
    mant[2, :, :200] = np.array([7, 11, 16, 32, 64, 65, 17, 18])[np.arange(200) % 8] & ((1 << bits[2, :, :200]) - 1)
    osc = rng.integers(0, 16, (n, 4 if joint else 1)).astype(np.int32)
    sw = rng.integers(0, 2, (n, nb)).astype(np.int32)
    for m16 in (False, True):
        out = {"overall_scale": _dev(torch, osc, torch.int32), "scale_factor": _dev(torch, sf, torch.int32),
               "bit_alloc": _dev(torch, ba, torch.int32),
               "mantissa": _dev(torch, mant.astype(np.uint16).view(np.int16) if m16 else mant, torch.int16 if m16 else torch.int32)}
        if joint:
            out["ms_switch"] = _dev(torch, sw, torch.int32)
        for use_huffman in (True, False):
            _check(torch, pacfile, enc, cfg, a, b, joint, out, use_huffman)
        forced = _dev(torch, rng.choice([0, 1, 2, 3, 15], (n, nch)), torch.int32)
        _check(torch, pacfile, enc, cfg, a, b, joint, out, True, given=forced)


def test_device_pack_reports_a_small_buffer_and_bad_tables(env):
    torch, pacfile, enc, cfg = env
    from mrcaudiocodec_amd import MrcError
    a = b = 1024
    nb = len(enc.h.bands(a, b))
    n = 8
    i32 = dict(dtype=torch.int32, device="cuda:0")
    osc, sf = torch.zeros((n, 1), **i32), torch.zeros((n, 1, nb), **i32)
    ba = torch.full((n, 1, nb), 8, **i32)
    mant = torch.full((n, 1, 1024), 3, **i32)
    buf = torch.full((4096,), 0xAA, dtype=torch.uint8, device="cuda:0")
    offs = torch.zeros((n + 1,), dtype=torch.int64, device="cuda:0")
    p = lambda t: t.data_ptr()
    with pytest.raises(MrcError):                                                # 8 raw-coded blocks need ~8 KB
        enc.h.dev_pack_blocks(a, b, n, 1, False, False, None, p(osc), None, p(sf), p(ba), p(mant), False, p(buf), 4096, p(offs))
    o = offs.cpu().numpy()
    total = int(o[n])
    assert total > 4096                                                         # the size law still reports what is needed
    fit = int(o[np.nonzero(o <= 4096)[0][-1]])                                   # end of the last chunk that fits
    assert bool((buf[fit:] == 0xAA).all())                                       # nothing written behind it
    bad = torch.full((n, 1), 7, **i32)
    big = torch.empty((total,), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(MrcError):
        enc.h.dev_pack_blocks(a, b, n, 1, False, True, p(bad), p(osc), None, p(sf), p(ba), p(mant), False, p(big), total, p(offs))
    assert enc.h.dev_pack_blocks(a, b, 0, 1, False, True, None, p(osc), None, p(sf), p(ba), p(mant), False, p(big), total, p(offs)) == 0


@pytest.mark.parametrize("case", ["a48", "b44"])
@pytest.mark.parametrize("which", ["_pac", "_pac_raw"])
def test_device_pack_reproduces_the_reference_cli_chunks(env, case, which):
    """the chunks of the reference's OWN .pac files (tests/golden/ref_pac.npz: block-switched stereo, with and without
    Huffman coding): parsed by the host parser, every block packed again ON THE DEVICE with the table ids the file
    carries -- the bytes must be the file's"""
    torch, pacfile, _, _ = env
    from mrcaudiocodec_amd.batch import StreamEncoder
    ref = G.load("ref_pac.npz")[case + which].tobytes()
    cfg, nch, _, off = pacfile.read_header(ref)
    enc = StreamEncoder(device_id=0, sample_rate=cfg.sample_rate)
    c = enc.h.cfg
    cfg.n_short, cfg.blksw_bits_a, cfg.blksw_bits_b = c.n_short, c.blksw_bits_a, c.blksw_bits_b
    chunks = pacfile.scan_chunks(ref, off)
    assert nch == 2 and len(chunks) % 2 == 0
    n_blocks = len(chunks) // 2
    ends = np.append(chunks[1:], len(ref))
    checked = 0
    for joint, sel in ((True, slice(0, 2 * (n_blocks - 1))), (False, slice(2 * (n_blocks - 1), None))):
        g = pacfile.unpack_blocks(cfg, ref, chunks[sel], 2, joint)
        first = chunks[sel][0::2]
        last_end = ends[sel][1::2]
        for i in range(len(g["a"])):
            a, b = int(g["a"][i]), int(g["b"][i])
            nb, half = len(enc.h.bands(a, b)), (a + b) // 2
            out = {"overall_scale": _dev(torch, g["overall_scale"][i:i + 1], torch.int32),
                   "scale_factor": _dev(torch, g["scale_factor"][i:i + 1, :, :nb], torch.int32),
                   "bit_alloc": _dev(torch, g["bit_alloc"][i:i + 1, :, :nb], torch.int32),
                   "mantissa": _dev(torch, g["mantissa"][i:i + 1, :, :half], torch.int32)}
            if joint:
                out["ms_switch"] = _dev(torch, g["ms_switch"][i:i + 1, :nb], torch.int32)
            table = _dev(torch, g["huff_table"][i:i + 1], torch.int32)
            got = enc.pack(a, b, out, use_huffman=True, huff_table=table)["bytes"].cpu().numpy().tobytes()
            assert got == ref[int(first[i]):int(last_end[i])], (joint, i, a, b)
            checked += 1
    assert checked == n_blocks


@pytest.mark.parametrize("joint", [False, True])
@pytest.mark.parametrize("use_huffman", [True, False])
def test_encode_stream_pcm16_pac_equals_encode_then_host_pack(env, joint, use_huffman):
    """host PCM -> host .pac chunk bytes in one pipelined call (9 chunks over the ring of 4 buffers, the last one short)
    == mrc_encode_stream_pcm16 followed by the host packer; too small a buffer is reported and retried"""
    torch, pacfile, enc, cfg = env
    from mrcaudiocodec_amd import MrcError
    F = 150
    rng = np.random.default_rng(11 + joint)
    level = np.repeat(10.0 ** rng.uniform(-3.5, -0.5, F + 1), 1024)
    pl = np.clip(np.round(rng.normal(0, 1, (F + 1) * 1024) * level * 32767), -32768, 32767).astype(np.int16)
    pr = np.clip(np.round(0.8 * pl + 0.2 * rng.normal(0, 1, pl.size) * level * 32767), -32768, 32767).astype(np.int16) if joint else None
    pl[:1024] = 0
    if joint:
        pr[:1024] = 0
    res_in = rng.integers(-50, 300, F).astype(np.int32)
    codes = enc.h.encode_stream_pcm16(pl, pr, res_in)
    if joint:
        want = pacfile.pack_joint_blocks(cfg, 1024, 1024, codes["overall_scale"], codes["ms_switch"], codes["scale_factor"],
                                         codes["bit_alloc"], codes["mantissa"], use_huffman)
    else:
        want = pacfile.pack_blocks(cfg, 1024, 1024, codes["overall_scale"], codes["scale_factor"], codes["bit_alloc"],
                                   codes["mantissa"], use_huffman)
    got = enc.h.encode_stream_pcm16_pac(pl, pr, res_in, use_huffman=use_huffman, chunk_frames=17)
    assert np.array_equal(got["block_offset"], want[1])
    assert np.array_equal(got["bytes"], want[0])
    assert np.array_equal(got["huff_table"], want[2])
    assert np.array_equal(got["bits_saved"], want[3])
    assert np.array_equal(got["reservoir_out"], codes["reservoir_out"])
    one = enc.h.encode_stream_pcm16_pac(pl, pr, res_in, use_huffman=use_huffman)          # default chunking: one chunk here
    assert np.array_equal(one["bytes"], want[0]) and np.array_equal(one["block_offset"], want[1])
    small = enc.h.encode_stream_pcm16_pac(pl, pr, res_in, use_huffman=use_huffman, chunk_frames=40, bytes_per_chunk=16)
    assert np.array_equal(small["bytes"], want[0])                                        # retried at the worst-case size
    import ctypes as C
    from mrcaudiocodec_amd import _lib
    tiny, offs, total = np.zeros(64, np.uint8), np.zeros(F + 1, np.int64), np.zeros(1, np.int64)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = _lib.lib.mrc_encode_stream_pcm16_pac(enc.h._h, F, vp(pl), vp(pr), vp(res_in), int(use_huffman), vp(tiny), tiny.size,
                                              vp(offs), None, None, None, total.ctypes.data_as(_lib._i64p), 17)
    assert rc == _lib.MRC_ERR_NOMEM and not tiny.any()                                    # reported, nothing written


def test_device_pack_with_other_field_widths():
    """field widths other than the defaults (3-bit scale factors, 2 + 0 block-switching bits): crafted codes, device == host"""
    import torch
    from mrcaudiocodec_amd import pacfile
    from mrcaudiocodec_amd.batch import StreamEncoder
    enc = StreamEncoder(device_id=0, n_scale_bits=3, blksw_bits_a=2, blksw_bits_b=0)
    c = enc.h.cfg
    cfg = pacfile.make_config(c.sample_rate, c.n_mdct_lines, c.n_short, c.n_scale_bits, c.n_mant_size_bits,
                              c.target_bits_per_sample, c.blksw_bits_a, c.blksw_bits_b)
    rng = np.random.default_rng(77)
    for (a, b) in ((1024, 1024), (128, 128)):
        lines = enc.h.bands(a, b)
        nb, half = len(lines), (a + b) // 2
        n = 40
        for joint in (False, True):
            nch = 2 if joint else 1
            ba = rng.integers(0, 17, (n, nch, nb)).astype(np.int32)
            ba[ba == 1] = 0
            sf = rng.integers(0, 8, (n, nch, nb)).astype(np.int32)
            bits = ba[:, :, np.repeat(np.arange(nb), lines)]
            mant = ((rng.integers(0, 1 << 16, bits.shape) % 40) & ((1 << bits) - 1)).astype(np.int32)
            out = {"overall_scale": _dev(torch, rng.integers(0, 8, (n, 4 if joint else 1)), torch.int32),
                   "scale_factor": _dev(torch, sf, torch.int32), "bit_alloc": _dev(torch, ba, torch.int32),
                   "mantissa": _dev(torch, mant, torch.int32)}
            if joint:
                out["ms_switch"] = _dev(torch, rng.integers(0, 2, (n, nb)), torch.int32)
            for use_huffman in (True, False):
                _check(torch, pacfile, enc, cfg, a, b, joint, out, use_huffman)


def test_stream_entry_points_on_empty_and_single_frame_streams(env):
    """n = 0 (only the prior hop) and n = 1 through both pipelined entry points, mono and stereo"""
    torch, pacfile, enc, cfg = env
    from oracle import fast
    rng = np.random.default_rng(3)
    for joint in (False, True):
        for n in (0, 1):
            pl = np.concatenate([np.zeros(1024), rng.normal(0, 3000, n * 1024)]).round().astype(np.int16)
            pr = np.concatenate([np.zeros(1024), rng.normal(0, 2000, n * 1024)]).round().astype(np.int16) if joint else None
            codes = enc.h.encode_stream_pcm16(pl, pr)
            pac = enc.h.encode_stream_pcm16_pac(pl, pr, use_huffman=True)
            assert codes["mantissa"].shape == (n, 2 if joint else 1, 1024)
            assert pac["block_offset"].shape == (n + 1,) and pac["block_offset"][0] == 0
            if n == 0:
                assert pac["bytes"].size == 0
                continue
            bl = np.array(fast.blocks_from_stream(G.pcm_to_float(pl), 1024))
            if joint:
                br = np.array(fast.blocks_from_stream(G.pcm_to_float(pr), 1024))
                want = fast.encode_joint_batch(bl, br, 1024, 1024)
                host = pacfile.pack_joint_blocks(cfg, 1024, 1024, codes["overall_scale"], codes["ms_switch"],
                                                 codes["scale_factor"], codes["bit_alloc"], codes["mantissa"], True)
            else:
                want = fast.encode_mono_batch(bl, 1024, 1024)
                host = pacfile.pack_blocks(cfg, 1024, 1024, codes["overall_scale"], codes["scale_factor"], codes["bit_alloc"],
                                           codes["mantissa"], True)
            assert np.array_equal(np.squeeze(codes["mantissa"]).astype(np.int64), np.squeeze(want["mantissa"]).astype(np.int64))
            assert np.array_equal(np.squeeze(codes["bit_alloc"]), np.squeeze(want["bit_alloc"]))
            assert np.array_equal(pac["bytes"], host[0]) and pac["block_offset"][1] == host[0].size
