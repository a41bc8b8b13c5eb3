"""
Transient detector / block-shape sequencer ("next" row f-2).  CPU part: the host logic
(mrcaudiocodec_amd.transient: threshold tests + look-ahead sequencing) against the oracle's restatement of
pacfileThem.py:1025-1056,1182-1214, fed with sub-block peaks computed by SciPy.  GPU part: the filtering
kernel (mrc_transient_peaks) against scipy.signal.sosfilt, and the whole chain on a stereo stream.
"""
import numpy as np
import pytest

from mrcaudiocodec_amd import synth, transient as ptr
from oracle import codec as ocodec, transient as otr


def _stereo_stream(n_hops):
    x, _ = synth.c4_transients(n_hops)
    y, _ = synth.c4_transients(n_hops, seed=7, period=7)
    return np.stack([x, 0.6 * x + 0.4 * y])


def _scipy_peaks(stream, sos, hop=1024, n_short=128):
    from scipy import signal
    n_hops = stream.shape[1] // hop - 1
    out = np.zeros((n_hops, stream.shape[0], hop // n_short + 1))
    for h in range(n_hops):
        for c in range(stream.shape[0]):
            y = np.abs(signal.sosfilt(sos, stream[c, (h + 1) * hop:(h + 2) * hop]))
            out[h, c, :-1] = y.reshape(-1, n_short).max(axis=1)
            out[h, c, -1] = y.max()
    return out


def test_sequencer_host_logic_matches_oracle():
    s = _stereo_stream(40)
    sos = otr.design_sos(48000)
    assert np.array_equal(sos, ptr.design_sos(48000))
    flags = ptr.transient_positions(_scipy_peaks(s, sos))
    got = ptr.shapes_from_flags(flags, 1024, 128)
    want = otr.block_shapes(s, ocodec.default_params(nChannels=2), sos)
    assert got == want
    kinds = {(a, b) for (_, a, b) in got}
    assert kinds == {(1024, 1024), (1024, 128), (128, 128), (128, 1024)}
    assert got[-1][0] + got[-1][1] + got[-1][2] == 40 * 1024        # hops 0..38 written, the last one is dropped


@pytest.mark.gpu
def test_transient_peaks_kernel_and_shapes_on_gpu():
    from mrcaudiocodec_amd import Handle
    h = Handle()
    try:
        s = _stereo_stream(64)
        sos = ptr.design_sos(48000)
        got = h.transient_peaks(s, sos)
        want = _scipy_peaks(s, sos)
        # same recurrence, same operation order as scipy's sosfilt: agreement to the last bits
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
        assert ptr.block_shapes(h, s, sos) == otr.block_shapes(s, ocodec.default_params(nChannels=2), sos)
        m = ptr.block_shapes(h, s[:1], sos)
        assert m == otr.block_shapes(s[:1], ocodec.default_params(nChannels=1), sos)
    finally:
        h.close()


# ------------------------------------------------------------------ WAV ingest + the encode CLI (next row f-3)
def _write_wav(path, pcm_lr, rate=48000):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(np.ascontiguousarray(pcm_lr.T.astype("<i2")).tobytes())


def _test_pcm(n):
    rng = np.random.default_rng(11)
    t = np.arange(n)
    left = 6000 * np.sin(2 * np.pi * 440 * t / 48000) + rng.normal(0, 300, n)
    left[5000:5100] += rng.normal(0, 15000, 100)                   # a click -> short blocks
    right = 0.7 * left + rng.normal(0, 200, n)
    pcm = np.clip(np.rint(np.stack([left, right])), -32768, 32767)
    pcm[0, 17] = -32768                                            # the code the reference maps to 0.0
    return pcm


def test_wav_ingest_matches_oracle(tmp_path):
    from mrcaudiocodec_amd import cli
    from oracle import pcmfile as opcm
    pcm = _test_pcm(7 * 1024 + 333)                                # partial last hop -> zero padded
    _write_wav(tmp_path / "a.wav", pcm)
    r1 = cli.read_wav(str(tmp_path / "a.wav"))
    r2 = opcm.read_wav(str(tmp_path / "a.wav"))
    assert r1[:3] == r2[:3] == (48000, 2, 7 * 1024 + 333)
    assert np.array_equal(r1[3], r2[3]) and r1[3].shape == (2, 8 * 1024) and r1[3][0, 17] == 0.0


@pytest.mark.gpu
def test_cli_encode_wav_bytes_equal_oracle(tmp_path):
    from mrcaudiocodec_amd import cli
    from oracle import pacfile as opac
    pcm = _test_pcm(9 * 1024 + 100)
    _write_wav(tmp_path / "b.wav", pcm)
    got = cli.encode_wav(str(tmp_path / "b.wav"), str(tmp_path / "b.pac"))
    want = opac.encode_wav(str(tmp_path / "b.wav"))
    assert got == want and (tmp_path / "b.pac").read_bytes() == want
    assert cli.encode_wav(str(tmp_path / "b.wav"), use_huffman=False) == opac.encode_wav(str(tmp_path / "b.wav"), huffman=False)


@pytest.mark.gpu
def test_detector_on_int16_codes_equals_detector_on_floats():
    # mrc_transient_peaks_ex: the WAV's int16 codes are converted on load exactly as pcmfile.py:91-100 does (-32768 -> 0.0)
    from mrcaudiocodec_amd import Handle, synth, transient as ptr
    h = Handle(device_id=0)
    try:
        pcm = _test_pcm(6 * 1024).astype(np.int16)
        codes = np.concatenate([np.zeros((2, 1024), np.int16), pcm], axis=1)
        sos = ptr.design_sos(48000)
        a = h.transient_peaks(codes, sos)
        b = h.transient_peaks(synth.pcm_to_float(codes), sos)
        assert np.array_equal(a, b)
        assert np.array_equal(ptr.block_shape_array(h, codes, sos), np.asarray(ptr.block_shapes(h, synth.pcm_to_float(codes), sos)))
    finally:
        h.close()


@pytest.mark.gpu
def test_detector_kernel_forms_agree_with_sosfilt():
    # the kernel has a specialised form (ten sections = the reference's cheby2(20, ...), hops on 16-byte boundaries,
    # coefficients in registers, samples eight / two per load) and a general one; both must give sosfilt's peaks --
    # for other section counts, and for a device stream that starts on an odd sample (dev entry point)
    import torch
    from scipy import signal
    from mrcaudiocodec_amd import Handle
    h = Handle(device_id=0)
    try:
        s = _stereo_stream(12)
        for order in (20, 6, 2, 32):
            sos = ptr.design_sos(48000) if order == 20 else signal.cheby2(order, 40, 9000.0 / 48000, "high", output="sos")
            got = h.transient_peaks(s, sos)
            want = _scipy_peaks(s, sos)
            assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), order
        sos = ptr.design_sos(48000)
        want = _scipy_peaks(s[:1], sos)
        for dtype, fmt in ((torch.float64, 0), (torch.int16, 1)):
            x = s[0] if fmt == 0 else np.clip(np.rint(s[0] * 32767), -32767, 32767).astype(np.int16)
            ref = _scipy_peaks(synth.pcm_to_float(x)[None], sos) if fmt else want
            buf = torch.zeros(len(x) + 8, dtype=dtype, device="cuda:0")
            for shift in (0, 1, 3):                                   # 0: aligned (specialised form); 1, 3: general form
                buf[shift:shift + len(x)] = torch.from_numpy(x).to("cuda:0")
                view = buf[shift:shift + len(x)]
                n_hops = len(x) // 1024 - 1
                peaks = torch.empty((n_hops, 1, 9), dtype=torch.float64, device="cuda:0")
                h.dev_transient_peaks(n_hops, 1, sos, view.data_ptr(), fmt, len(x), peaks.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert np.abs(peaks.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max(), (fmt, shift)
    finally:
        h.close()
