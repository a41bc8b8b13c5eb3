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
