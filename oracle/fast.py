"""
oracle.fast -- batched / vectorised flavour of the encode-path oracle (TEST ORACLE).

Same arithmetic, same operation and summation orders as the faithful one-block functions in
oracle.{window,mdct,psychoac,quantize,bitalloc,ms_stereo,codec} (which cite the reference lines);
only the redundancy is removed: window tables are built once per shape, each masked threshold is
evaluated once, the peak loop runs over all frames of a batch at the same time.  tests/test_oracle.py
checks fast == faithful.  Used by the GPU parity tests so that 10^3..10^4 frames finish in seconds.

All frames of one call share the block shape (a, b).  Outputs are DENSE: the mantissa plane has
halfN entries per frame with 0 where the band got no bits (the reference omits those bands;
`compact_mantissa` rebuilds its layout).
"""
import numpy as np

from . import window as _w
from .psychoac import (Thresh, Bark, Intensity, SPL, ScaleFactorBands, AssignMDCTLinesFromFreqLimits,
                       shortFreqLimits, py2div)
from .bitalloc import BitAlloc

_kbd_cache = {}


def transition_table(a, b):
    """window.py:104-121 as a multiplicative table: rising half of KBD(2a) then falling half of KBD(2b)."""
    for n in (2 * a, 2 * b):
        if n not in _kbd_cache:
            _kbd_cache[n] = _w.kbd_table(n)
    return np.append(_kbd_cache[2 * a][:a], _kbd_cache[2 * b][b:])


def blocks_from_stream(x, hop, n_frames=None):
    """pacfileThem.py:628-631: block i = hop i-1 (prior) || hop i, for a stream that already starts with
    the prior hop (zeros at file start).  x: [..., (n_frames+1)*hop] -> [..., n_frames, 2*hop] (a view)."""
    x = np.asarray(x)
    total = x.shape[-1] // hop - 1
    n_frames = total if n_frames is None else n_frames
    s = x.strides[-1]
    return np.lib.stride_tricks.as_strided(x, shape=x.shape[:-1] + (n_frames, 2 * hop),
                                           strides=x.strides[:-1] + (hop * s, s), writeable=False)


def bands_for(a, b, nMDCTLines=1024, sampleRate=48000):
    half = (a + b) // 2
    if a + b == 2 * nMDCTLines:
        return ScaleFactorBands(AssignMDCTLinesFromFreqLimits(half, sampleRate))
    return ScaleFactorBands(AssignMDCTLinesFromFreqLimits(half, sampleRate, shortFreqLimits))


def mdct_batch(blocks, a, b):
    """Windowed forward MDCT of every row (window.py:104-121 + mdct.py:63-76). -> [B, (a+b)/2]."""
    blocks = np.asarray(blocks, dtype=np.float64)
    N = a + b
    xw = np.multiply(blocks, transition_table(a, b))
    n = np.arange(N)
    n0 = (b + 1.0) / 2.0
    pre = np.exp(np.multiply(n, -1j * np.pi / N))
    spec = np.fft.fft(np.multiply(pre, xw), N, axis=-1)
    k = np.add(np.arange(N // 2), 1.0 / 2.0)
    post = np.exp(np.multiply(k, -1j * 2.0 * np.pi * n0 / N))
    return (2.0 / N) * np.real(np.multiply(post, spec[..., 0:N // 2]))


def _mag_code(v, nBits):
    """|code| of quantize.py:12-38 / 61-87 for v >= 0, nBits broadcastable (int array)."""
    nBits = np.asarray(nBits)
    c = np.power(2.0, nBits) - 1.0
    clip = np.power(2.0, nBits - 1) - 1.0
    t = ((c * v) + 1.0) / 2.0
    return np.where(v >= 1.0, clip, np.trunc(t)).astype(np.int64)


def _floor_log2(code):
    """floor(log2(code)) for positive int64 (exact); 0 for code == 0 (quantize.py:133-137)."""
    m, e = np.frexp(code.astype(np.float64))       # exact for codes < 2^53
    return np.where(code > 0, e - 1, 0).astype(np.int64)


def scale_factor_batch(v, nScaleBits, nMantBits):
    """quantize.py:114-146 for arrays: v >= 0 (a max of absolute values), nMantBits broadcastable."""
    v = np.asarray(v, dtype=np.float64)
    nBits = (1 << nScaleBits) - 1 + np.asarray(nMantBits, dtype=np.int64)
    code = _mag_code(v, nBits)
    lz = (nBits - 2) - _floor_log2(code)
    return np.minimum(lz, (1 << nScaleBits) - 1).astype(np.int64)


def mantissa_batch(x, scale, nScaleBits, nMantBits):
    """quantize.py:294-322 for arrays; scale / nMantBits broadcast against x.  int64 result."""
    x = np.asarray(x, dtype=np.float64)
    nMantBits = np.asarray(nMantBits, dtype=np.int64)
    scale = np.asarray(scale, dtype=np.int64)
    cap = (1 << nScaleBits) - 1
    nBits = cap + nMantBits
    mag = _mag_code(np.abs(x), nBits)
    mag = np.where(x == 0.0, 0, mag)
    shifted = np.right_shift(mag, np.maximum(cap - scale, 0))
    signbit = np.where(x < 0.0, np.left_shift(np.int64(1), np.maximum(nMantBits - 1, 0)), 0)
    return signbit + shifted


def masked_threshold_batch(blocks, halfN, sampleRate):
    """psychoac.py:134-173 for every row of blocks[B, N] at once -> [B, halfN] (dB SPL)."""
    blocks = np.asarray(blocks, dtype=np.float64)
    B, N = blocks.shape
    n = np.arange(halfN)
    MDCTFreq = (n + 0.5) * ((float(sampleRate) / halfN) / 2.)
    X = np.fft.fft(np.multiply(blocks, _hann(N)), axis=-1)      # window.py:28-45: data * table
    XI = 4. * (np.abs(X) ** 2.) / ((N ** 2.) * (3. / 8.))
    total = np.tile(Intensity(Thresh(MDCTFreq)), (B, 1))
    zb = Bark(MDCTFreq)
    binHz = py2div(sampleRate, N)
    last = N // 2 - 100                              # loop variable i runs 2 .. last-1, peak bin p = i-1
    c = XI[:, 1:last - 1]                            # p = 1 .. last-2
    is_peak = (c > XI[:, 0:last - 2]) & (c > XI[:, 2:last])
    counts = is_peak.sum(axis=1)
    Pmax = int(counts.max()) if B else 0
    # compact peak bins per row, in increasing bin order
    order = np.argsort(~is_peak, axis=1, kind="stable")[:, :Pmax] + 1 if Pmax else np.zeros((B, 0), dtype=int)
    rows = np.arange(B)[:, None]
    p = order
    x0 = XI[rows, p - 1]; x1 = XI[rows, p]; x2 = XI[rows, p + 1]
    s3 = (x0 + x1) + x2
    with np.errstate(invalid="ignore", divide="ignore"):
        level = SPL(s3)
        f = binHz * (((p - 1) * x0 + p * x1) + (p + 1) * x2) / s3
        zm = Bark(f)
        boost = 0.37 * np.maximum(level - 40, 0)
    for j in range(Pmax):
        act = np.nonzero(counts > j)[0]
        dz = zb[None, :] - zm[act, j][:, None]
        adz = np.abs(dz)
        above = dz > 0.5
        outside = adz > 0.5
        total[act] += Intensity(level[act, j][:, None] - 15.0
                                + -27 * (adz - 0.5) * outside
                                + boost[act, j][:, None] * (adz - 0.5) * outside * above)
    return SPL(total)


_hann_cache = {}


def _hann(N):
    if N not in _hann_cache:
        _hann_cache[N] = _w.HanningWindow(np.ones(N))
    return _hann_cache[N]


def smr_batch(blocks, scaled_lines, overall_scale, sampleRate, sfBands):
    """psychoac.py:176-219 per row: SMR[b] = max over band of (SPL(4 X^2) - 6 scale - masked threshold)."""
    thr = masked_threshold_batch(blocks, scaled_lines.shape[1], sampleRate)
    spl = SPL(2. * (np.abs(scaled_lines) ** 2.) / (1. / 2.)) - 6. * np.asarray(overall_scale)[:, None]
    excess = spl - thr
    out = np.empty((blocks.shape[0], sfBands.nBands))
    for i in range(sfBands.nBands):
        out[:, i] = np.amax(excess[:, sfBands.lowerLine[i]:sfBands.upperLine[i] + 1], axis=1)
    return out


def overall_scale_batch(lines, nScaleBits):
    """codecThem.py:321-323: ScaleFactor(max|X|, nScaleBits) (nMantBits default 5); returns (scale, X * 2^scale)."""
    s = scale_factor_batch(np.max(np.abs(lines), axis=1), nScaleBits, 5)
    return s, lines * np.power(2.0, s)[:, None]


def _quantise_batch(lines, bitAlloc, sfBands, nScaleBits):
    """codecThem.py:335-350 for all rows: (scaleFactor[B,nBands], dense mantissa[B,halfN])."""
    B, halfN = lines.shape
    sf = np.empty((B, sfBands.nBands), dtype=np.int64)
    mant = np.zeros((B, halfN), dtype=np.int64)
    for i in range(sfBands.nBands):
        lo, hi = sfBands.lowerLine[i], sfBands.upperLine[i] + 1
        ba = bitAlloc[:, i]
        sf[:, i] = scale_factor_batch(np.max(np.abs(lines[:, lo:hi]), axis=1), nScaleBits, ba)
        m = mantissa_batch(lines[:, lo:hi], sf[:, i][:, None], nScaleBits, ba[:, None])
        mant[:, lo:hi] = np.where(ba[:, None] > 0, m, 0)
    return sf, mant


def mono_budget(cp_like, halfN, nBands):
    """codecThem.py:299-306 without the reservoir term (added last, line 308)."""
    b = cp_like["targetBitsPerSample"] * float(halfN)
    b -= cp_like["nScaleBits"] * (nBands + 1)
    b -= cp_like["nMantSizeBits"] * nBands
    b -= cp_like["blkswBitA"]
    b -= cp_like["blkswBitB"]
    return b


def joint_budget(cp_like, halfN, nBands, reservoir):
    """codecThem.py:381-396 (the reservoir enters before the block-switch bits)."""
    b = cp_like["targetBitsPerSample"] * float(halfN)
    b -= cp_like["nScaleBits"] * nBands
    b -= cp_like["nMantSizeBits"] * nBands
    b += b
    b -= nBands
    b -= cp_like["nScaleBits"] * 4
    b += reservoir
    b -= cp_like["blkswBitA"]
    b -= cp_like["blkswBitB"]
    return b


DEFAULTS = dict(sampleRate=48000, nMDCTLines=1024, nScaleBits=4, nMantSizeBits=4,
                targetBitsPerSample=2.86, blkswBitA=1, blkswBitB=1)


def encode_mono_batch(blocks, a, b, reservoir_in=None, params=None):
    """codecThem.py:281-354 for every row of blocks[B, a+b], each with its own incoming reservoir
    (independent-frames mode).  Returns a dict of arrays."""
    P = dict(DEFAULTS, **(params or {}))
    blocks = np.asarray(blocks, dtype=np.float64)
    B = blocks.shape[0]
    halfN = (a + b) // 2
    sfb = bands_for(a, b, P["nMDCTLines"], P["sampleRate"])
    reservoir_in = np.zeros(B, dtype=np.int64) if reservoir_in is None else np.asarray(reservoir_in)
    maxMant = min(16, 1 << P["nMantSizeBits"])
    X = mdct_batch(blocks, a, b)
    scale, Xs = overall_scale_batch(X, P["nScaleBits"])
    smr = smr_batch(blocks, Xs, scale, P["sampleRate"], sfb)
    base = mono_budget(P, halfN, sfb.nBands)
    ba = np.empty((B, sfb.nBands), dtype=np.int64)
    res = np.empty(B, dtype=np.int64)
    for i in range(B):
        bits, left = BitAlloc(base + int(reservoir_in[i]), maxMant, sfb.nBands, sfb.nLines, smr[i].copy())
        ba[i] = bits.astype(int)
        res[i] = int(left)
    sf, mant = _quantise_batch(Xs, ba, sfb, P["nScaleBits"])
    return dict(mdct=X, overall_scale=scale, smr=smr, bit_alloc=ba, scale_factor=sf, mantissa=mant,
                reservoir_out=res, sfBands=sfb)


def ms_switch_batch(XL, XR, sfb):
    """ms_stereo.py:5-27 per row (np.sum per slice keeps NumPy's pairwise summation order)."""
    l2 = np.square(XL); r2 = np.square(XR)
    d = np.abs(l2 - r2); s = np.abs(l2 + r2)
    out = np.empty((XL.shape[0], sfb.nBands), dtype=np.int64)
    for i in range(sfb.nBands):
        lo, hi = sfb.lowerLine[i], sfb.upperLine[i] + 1
        for r in range(XL.shape[0]):
            out[r, i] = 1 if np.sum(d[r, lo:hi]) < 0.8 * np.sum(s[r, lo:hi]) else 0
    return out


def encode_joint_batch(left, right, a, b, reservoir_in=None, params=None):
    """codecThem.py:359-574 for every row pair.  overall_scale columns are L, R, M, S."""
    P = dict(DEFAULTS, **(params or {}))
    left = np.asarray(left, dtype=np.float64)
    right = np.asarray(right, dtype=np.float64)
    B = left.shape[0]
    halfN = (a + b) // 2
    sfb = bands_for(a, b, P["nMDCTLines"], P["sampleRate"])
    nb = sfb.nBands
    reservoir_in = np.zeros(B, dtype=np.int64) if reservoir_in is None else np.asarray(reservoir_in)
    maxMant = min(16, 1 << P["nMantSizeBits"])
    time = [left, right, (left + right) / 2.0, (left - right) / 2.0]
    X = [mdct_batch(t, a, b) for t in time]
    sw = ms_switch_batch(X[0], X[1], sfb)
    scale, Xs, smr = [], [], []
    for t, x in zip(time, X):
        s, xs = overall_scale_batch(x, P["nScaleBits"])
        scale.append(s); Xs.append(xs)
        smr.append(smr_batch(t, xs, s, P["sampleRate"], sfb))
    on = sw == 1
    smr1 = np.where(on, smr[2], smr[0])
    smr2 = np.where(on, smr[3], smr[1])
    nLinesPass = np.append(sfb.nLines, sfb.nLines)
    ba = np.empty((B, 2 * nb), dtype=np.int64)
    res = np.empty(B, dtype=np.int64)
    for i in range(B):
        bits, leftover = BitAlloc(joint_budget(P, halfN, nb, int(reservoir_in[i])), maxMant, 2 * nb,
                                  nLinesPass, np.append(smr1[i], smr2[i]))
        ba[i] = bits.astype(int)
        res[i] = int(leftover)
    band_of_line = np.repeat(np.arange(nb), sfb.nLines)
    on_line = on[:, band_of_line]
    lines1 = np.where(on_line, Xs[2], Xs[0])
    lines2 = np.where(on_line, Xs[3], Xs[1])
    sf1, m1 = _quantise_batch(lines1, ba[:, :nb], sfb, P["nScaleBits"])
    sf2, m2 = _quantise_batch(lines2, ba[:, nb:], sfb, P["nScaleBits"])
    return dict(mdct=np.stack(X, axis=1), overall_scale=np.stack(scale, axis=1), ms_switch=sw,
                smr=np.stack(smr, axis=1), bit_alloc=np.stack([ba[:, :nb], ba[:, nb:]], axis=1),
                scale_factor=np.stack([sf1, sf2], axis=1), mantissa=np.stack([m1, m2], axis=1),
                reservoir_out=res, sfBands=sfb)


def compact_mantissa(dense_row, bit_alloc_row, sfBands):
    """Dense [halfN] plane -> the reference's layout (bands with 0 bits omitted), int32."""
    keep = np.repeat(np.asarray(bit_alloc_row) > 0, sfBands.nLines)
    return np.asarray(dense_row)[keep].astype(np.int32)
