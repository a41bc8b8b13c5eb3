"""
oracle.transient -- transient detector and block-shape sequencing of the encoder CLI (TEST ORACLE).

Restates pacfileThem.py:1025-1056 (TransientDetector), 1146-1154 (filter design and thresholds) and the
block-switching decisions of the driver loop, pacfileThem.py:1159-1214.  Uses SciPy for the filter design
and the filtering exactly like the reference (signal.cheby2 / tf2sos / sosfilt).  Pinned end to end by
tests/golden/ref_pac.npz: the .pac files the reference's own driver wrote for WAV files with bursts contain its
block-shape decisions, and the oracle's bytes equal them (tests/test_reference_golden.py).
"""
import numpy as np
from scipy import signal

from .psychoac import py2div

THRESHOLDS = np.array([0.1, 0.075])              # pacfileThem.py:1154


def design_sos(sampleRate):
    """pacfileThem.py:1146-1147: 20th-order Chebyshev-II high-pass, 40 dB, Wn = 9000/sampleRate, as SOS."""
    b, a = signal.cheby2(20, 40, 9000. / sampleRate, 'high')
    return signal.tf2sos(b, a)


def TransientDetector(data, codingParams, sos, T):
    """pacfileThem.py:1025-1056.  data: [nChannels][nSamplesPerBlock].  Updates codingParams.P in place.
    Every hop is filtered from a ZERO filter state (sosfilt is called without zi)."""
    cp = codingParams
    nSub = py2div(cp.nSamplesPerBlock, cp.nSamplesShort)
    blksw = np.array([])
    for iCh in range(cp.nChannels):
        dataFilt = signal.sosfilt(sos, data[iCh])
        for i in range(nSub):
            cp.P[iCh][i + 1] = np.amax(np.abs(dataFilt[i * cp.nSamplesShort:(i + 1) * cp.nSamplesShort]))
        if np.amax(np.abs(dataFilt)) > T[0]:
            for i in range(nSub):
                if cp.P[iCh][i + 1] * T[1] > cp.P[iCh][i]:
                    blksw = np.append(blksw, i + 1)
    cp.P[:, 0] = cp.P[:, nSub]
    return np.unique(blksw[np.nonzero(blksw)])


def block_shapes(stream, cp, sos=None, T=THRESHOLDS):
    """The (offset, a, b) sequence the CLI's encode loop produces (pacfileThem.py:1159-1214) for a stream
    [nChannels][(nHops+1)*hop] that starts with the zero prior hop.  One hop of look-ahead: hop i is written
    when hop i+1 has been analysed, as 8 short blocks if sum(blksw_i) > 1 or any(blksw_{i+1} == 1), else as
    one long block; the LAST hop is never written (the loop ends before it; Close() only flushes zeros)."""
    stream = np.asarray(stream, dtype=np.float64)
    hop = cp.nSamplesPerBlock
    nHops = stream.shape[1] // hop - 1
    sos = design_sos(cp.sampleRate) if sos is None else sos
    nSub = py2div(hop, cp.nSamplesShort)
    cp.P = np.zeros((cp.nChannels, 1 + nSub))
    shapes = []
    off, a = 0, hop
    mem = None
    for i in range(nHops):
        data = stream[:, (i + 1) * hop:(i + 2) * hop]
        blksw = TransientDetector(data, cp, sos, T)
        if mem is not None:
            if np.sum(mem) > 1 or np.any(blksw == 1):
                for _ in range(nSub):
                    shapes.append((off, a, cp.nSamplesShort)); off += a; a = cp.nSamplesShort
            else:
                shapes.append((off, a, hop)); off += a; a = hop
        mem = blksw
    return shapes
