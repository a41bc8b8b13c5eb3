"""
Drop-in replacement for the encode half of the reference's `codecThem` module: same function names,
argument meaning, return shapes and side effects on `codingParams`, computed by the gfx950 kernels
behind libmrc_hip.so.  pacfileThem.py:108 does `import codecThem as codec` and calls
codec.Encode / codec.JointEncode (pacfileThem.py:987-1003); a maintainer switches with

    import mrcaudiocodec_amd.codecThem as codec

Kept from the reference:
  EncodeSingleChannel(data, codingParams)        codecThem.py:281-354
  JointEncodeChannels(dataLeft, dataRight, cp)   codecThem.py:359-574
  Encode / EncodeNoHuff / JointEncode            codecThem.py:205-278
  calculateHuffmanGain(mantissa, bitAlloc, cp)   codecThem.py:136-203  (host side, as BASELINE.json's north_star says)
  Decode / JointDecode                           codecThem.py:30-134   (the decoder core; "next" row f-4)
  L1 names re-exported by codecThem.py:14-21:    TransitionWindow, KBDWindow, MDCT, CalcSMRs, getMaskedThreshold, Bark,
      BitAlloc, ScaleFactor, QuantizeUniform, vQuantizeUniform, Mantissa, vMantissa, MSSwitchSFBands,
      StereoMaskingFactor, OverallSMRs -- each runs on the GPU (OverallSMRs is a per-band select on the host).
codingParams is the reference's attribute bag (audiofile.py:51-53): read a, b, nMDCTLines, nScaleBits,
nMantSizeBits, targetBitsPerSample, sampleRate, sfBands, blkswBitA/B, nChannels, bitReservoir;
written bitReservoir (codecThem.py:224,274,332,503).
Differences, all documented in DESIGN.md: Huffman table ids follow sorted table names instead of
directory order; tables come from package data, not ./training_data/*.pkl.  There is no CPU fallback.
"""
import os

import numpy as np

from . import _lib
from .huffman_tables import CODES, ESCAPE, RAW_TABLE_ID, TABLE_NAMES

_handles = {}


def _handle(cp):
    key = (int(cp.sampleRate), int(getattr(cp, "nMDCTLines", 1024)), int(getattr(cp, "nSamplesShort", 128)),
           int(cp.nScaleBits), int(cp.nMantSizeBits), float(getattr(cp, "targetBitsPerSample", 2.86)),   # decode: unset
           int(getattr(cp, "blkswBitA", 0)), int(getattr(cp, "blkswBitB", 0)),
           int(getattr(cp, "deviceId", os.environ.get("MRC_DEVICE", 0))))
    if not isinstance(cp.sampleRate, (int, np.integer)):
        raise TypeError("codingParams.sampleRate must be an integer (the reference's sampleRate/N is an integer "
                        "division, psychoac.py:165)")
    h = _handles.get(key)
    if h is None:
        h = _lib.Handle(sample_rate=key[0], n_mdct_lines=key[1], n_short=key[2], n_scale_bits=key[3],
                        n_mant_size_bits=key[4], target_bits_per_sample=key[5], blksw_bits_a=key[6],
                        blksw_bits_b=key[7], device_id=key[8])
        _handles[key] = h
    return h


def _bands_checked(h, cp):
    a, b = int(cp.a), int(cp.b)
    nl = h.bands(a, b)
    got = np.asarray(cp.sfBands.nLines)
    if got.shape != nl.shape or not np.array_equal(got, nl):
        raise ValueError("codingParams.sfBands differs from the band table of block shape (%d,%d) at %d Hz "
                         "(pacfileThem.py:637-645); custom band tables are not supported" % (a, b, cp.sampleRate))
    return a, b, nl


def _compact(dense, bit_alloc, n_lines):
    # the reference transmits no mantissas for bands with 0 bits (codecThem.py:336-340)
    return np.ascontiguousarray(dense[np.repeat(np.asarray(bit_alloc) > 0, n_lines)], dtype=np.int32)


# ------------------------------------------------------------------------------------------------ L2
def EncodeSingleChannel(data, codingParams):
    """codecThem.py:281-354 -> (scaleFactor int32[nBands], bitAlloc int[nBands], mantissa int32[compact], overallScale)."""
    h = _handle(codingParams)
    a, b, nl = _bands_checked(h, codingParams)
    r = h.encode_mono(np.asarray(data, dtype=np.float64)[None, :], a, b, [int(codingParams.bitReservoir)])
    bitAlloc = r["bit_alloc"][0].astype(int)
    codingParams.bitReservoir = int(r["reservoir_out"][0])
    return (r["scale_factor"][0].copy(), bitAlloc, _compact(r["mantissa"][0], bitAlloc, nl), int(r["overall_scale"][0]))


def JointEncodeChannels(dataLeft, dataRight, codingParams):
    """codecThem.py:359-574 -> ([sf1,sf2],[ba1,ba2],[m1,m2],[osL,osR,osM,osS], ms_switch list)."""
    h = _handle(codingParams)
    a, b, nl = _bands_checked(h, codingParams)
    r = h.encode_joint(np.asarray(dataLeft, dtype=np.float64)[None, :], np.asarray(dataRight, dtype=np.float64)[None, :],
                       a, b, [int(codingParams.bitReservoir)])
    ba = [r["bit_alloc"][0, c].astype(int) for c in range(2)]
    codingParams.bitReservoir = int(r["reservoir_out"][0])
    return ([r["scale_factor"][0, c].copy() for c in range(2)], ba,
            [_compact(r["mantissa"][0, c], ba[c], nl) for c in range(2)],
            [int(v) for v in r["overall_scale"][0]], [int(v) for v in r["ms_switch"][0]])


_LEN_LUT = {}


def _length_lut(name):
    """code length per mantissa value (0 = not in table) for values < 2^17."""
    if name not in _LEN_LUT:
        lut = np.zeros(1 << 17, dtype=np.int64)
        for v, code in CODES[name].items():
            lut[v] = len(code)
        _LEN_LUT[name] = lut
    return _LEN_LUT[name]


def calculateHuffmanGain(mantissa, bitAlloc, codingParams):
    """codecThem.py:136-203 on the host: pick the table with the fewest bits (first wins ties, must beat raw),
    return (table id | 15, mantissas | list of code strings "0101" / "<escape>/<value>", bits_saved).
    The reference prices the escape VALUE itself as its code alone (169-172) although it emits
    escape+raw (194-200); that under-count is kept."""
    sfBands = codingParams.sfBands
    bitAlloc = np.asarray(bitAlloc)
    nLines = np.asarray(sfBands.nLines)
    on = bitAlloc > 0
    raw_bits = int(np.sum(bitAlloc[on] * nLines[on]))
    m = np.asarray(mantissa, dtype=np.int64)
    ba_line = np.repeat(bitAlloc[on], nLines[on])
    best, table_to_use = raw_bits, RAW_TABLE_ID
    for i, name in enumerate(TABLE_NAMES):
        lut = _length_lut(name)
        ln = lut[m]
        cost = int(np.sum(np.where(ln > 0, ln, ba_line + lut[ESCAPE[name]])))
        if cost < best:
            best, table_to_use = cost, i
    if table_to_use == RAW_TABLE_ID:
        codes = mantissa
    else:
        name = TABLE_NAMES[table_to_use]
        table, esc = CODES[name], ESCAPE[name]
        codes = [table[v] if (v in table and v != esc) else table[esc] + "/" + str(v) for v in m.tolist()]
    return (table_to_use, codes, raw_bits - best)


def Encode(data, codingParams):
    """codecThem.py:205-231."""
    scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable = [], [], [], [], []
    for iCh in range(codingParams.nChannels):
        (s, b, m, o) = EncodeSingleChannel(data[iCh], codingParams)
        (table_to_use, new_m, bits_saved) = calculateHuffmanGain(m, b, codingParams)
        codingParams.bitReservoir += bits_saved
        scaleFactor.append(s); bitAlloc.append(b); mantissa.append(new_m)
        overallScaleFactor.append(o); huffTable.append(table_to_use)
    return (scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable)


def EncodeNoHuff(data, codingParams):
    """codecThem.py:234-260."""
    scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable = [], [], [], [], []
    for iCh in range(codingParams.nChannels):
        (s, b, m, o) = EncodeSingleChannel(data[iCh], codingParams)
        scaleFactor.append(s); bitAlloc.append(b); mantissa.append(m)
        overallScaleFactor.append(o); huffTable.append(RAW_TABLE_ID)
    return (scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable)


def JointEncode(data, codingParams):
    """codecThem.py:262-278."""
    (scaleFactor, bitAlloc, mantissa, overallScaleFactor, ms_switch) = \
        JointEncodeChannels(data[0], data[1], codingParams)
    new_mantissa, huffTable = [], []
    for iCh in range(codingParams.nChannels):
        (table_to_use, new_m, bits_saved) = calculateHuffmanGain(mantissa[iCh], bitAlloc[iCh], codingParams)
        codingParams.bitReservoir += bits_saved
        huffTable.append(table_to_use)
        new_mantissa.append(new_m)
    return (scaleFactor, bitAlloc, new_mantissa, overallScaleFactor, ms_switch, huffTable)


# ------------------------------------------------------------------------------------------------ L1
# The reference's helper functions keep their signatures; they carry no codingParams, so they run on a
# default-parameter handle (48 kHz unless sampleRate is an argument).
def _dense(mantissa, n_lines):
    """The decoder's mantissa array (pacfileThem.py:221) holds N/2 or more entries, dense."""
    half = int(np.sum(n_lines))
    m = np.zeros(half, dtype=np.int32)
    src = np.asarray(mantissa, dtype=np.int64)[:half]
    m[:len(src)] = src
    return m


def Decode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, codingParams):
    """codecThem.py:30-63: one channel's windowed block of a+b samples (before overlap-and-add)."""
    h = _handle(codingParams)
    _, _, n_lines = _bands_checked(h, codingParams)
    out = h.decode(codingParams.a, codingParams.b, [int(overallScaleFactor)], np.asarray(scaleFactor)[None, None, :],
                   np.asarray(bitAlloc)[None, None, :], _dense(mantissa, n_lines)[None, None, :])
    return out[0, 0]


def JointDecode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, codingParams, ms_switch):
    """codecThem.py:65-134: [left, right] windowed blocks; overallScaleFactor = [L, R, M, S]."""
    h = _handle(codingParams)
    _, _, n_lines = _bands_checked(h, codingParams)
    out = h.decode(codingParams.a, codingParams.b, np.asarray(overallScaleFactor, dtype=np.int32)[None, :],
                   np.stack([np.asarray(scaleFactor[0]), np.asarray(scaleFactor[1])])[None],
                   np.stack([np.asarray(bitAlloc[0]), np.asarray(bitAlloc[1])])[None],
                   np.stack([_dense(mantissa[0], n_lines), _dense(mantissa[1], n_lines)])[None],
                   np.asarray(ms_switch, dtype=np.int32)[None, :])
    return [out[0, 0], out[0, 1]]


def _default_handle(sampleRate=48000):
    class _P:
        pass
    p = _P()
    p.sampleRate, p.nMDCTLines, p.nSamplesShort, p.nScaleBits, p.nMantSizeBits = int(sampleRate), 1024, 128, 4, 4
    p.targetBitsPerSample, p.blkswBitA, p.blkswBitB = 2.86, 1, 1
    return _handle(p)


def TransitionWindow(dataSampleArray, a, b):
    """window.py:104-121."""
    x = np.asarray(dataSampleArray, dtype=np.float64)
    return _default_handle().window(x[None, :], int(a), int(b))[0]


def KBDWindow(dataSampleArray, alpha=4.):
    """window.py:49-101 (alpha is fixed at 4, the only value the codec uses)."""
    if alpha != 4.:
        raise ValueError("only alpha = 4 is supported")
    n = np.size(dataSampleArray)
    return TransitionWindow(dataSampleArray, n // 2, n // 2)


def MDCT(data, a, b, isInverse=False):
    """mdct.py:63-76 (forward only; the caller windows first, as in codecThem.py:316-317)."""
    if isInverse:
        raise NotImplementedError("the inverse transform is decode-side and out of scope")
    return _default_handle().mdct(np.asarray(data, dtype=np.float64)[None, :], int(a), int(b), apply_window=False)[0][0]


def _shape_of(n_samples, sfBands, h):
    # CalcSMRs / getMaskedThreshold get no (a,b); the threshold only depends on N and the band table
    for (a, b) in ((1024, 1024), (128, 128), (1024, 128)):
        if a + b == n_samples:
            return a, b
    raise ValueError("unsupported block length %d" % n_samples)


def CalcSMRs(data, MDCTdata, MDCTscale, sampleRate, sfBands, ms=0, preCalcThresh=0.0):
    """psychoac.py:176-219 (ms / preCalcThresh have no effect in the reference either: line 210)."""
    h = _default_handle(sampleRate)
    x = np.asarray(data, dtype=np.float64)
    a, b = _shape_of(x.size, sfBands, h)
    return h.smr(x[None, :], a, b, np.asarray(MDCTdata, dtype=np.float64)[None, :], [int(MDCTscale)])[0]


def getMaskedThreshold(data, MDCTdata, MDCTscale, sampleRate, sfBands):
    """psychoac.py:134-173."""
    h = _default_handle(sampleRate)
    x = np.asarray(data, dtype=np.float64)
    a, b = _shape_of(x.size, sfBands, h)
    return h.smr(x[None, :], a, b, np.asarray(MDCTdata, dtype=np.float64)[None, :], [int(MDCTscale)],
                 want_thresh=True)[1][0]


def BitAlloc(bitBudget, maxMantBits, nBands, nLines, SMR):
    """bitalloc.py:106-155 -> (bits float64[nBands], int(bitsLeft)).  Like the reference, a float64 NumPy SMR array is
    UPDATED IN PLACE (bitalloc.py:132-151): -12 for a band's first grant (2 bits), -6 for every further bit,
    -99999999999999999.0 once the band is retired -- the running values come back from the kernel."""
    bits, left, after = _default_handle().bitalloc(float(bitBudget), int(maxMantBits), np.asarray(nLines)[:nBands],
                                                   np.asarray(SMR, dtype=np.float64)[:nBands], want_smr_after=True)
    if isinstance(SMR, np.ndarray) and SMR.dtype == np.float64:
        SMR[:nBands] = after[0]
    return (bits[0].astype(np.float64), int(left[0]))


def ScaleFactor(aNum, nScaleBits=3, nMantBits=5):
    """quantize.py:114-146."""
    return int(_default_handle().scale_factor([float(aNum)], nScaleBits, nMantBits)[0])


def vMantissa(aNumVec, scale, nScaleBits=3, nMantBits=5):
    """quantize.py:294-322 (float64 integer-valued result, like the reference)."""
    return _default_handle().mantissa(aNumVec, int(scale), nScaleBits, int(nMantBits)).astype(np.float64)


def Mantissa(aNum, scale, nScaleBits=3, nMantBits=5):
    """quantize.py:222-249: the scalar form."""
    return int(_default_handle().mantissa([float(aNum)], int(scale), nScaleBits, int(nMantBits))[0])


def vQuantizeUniform(aNumVec, nBits):
    """quantize.py:61-87 (float64 integer-valued result, like the reference)."""
    return _default_handle().quantize_uniform(np.asarray(aNumVec, dtype=np.float64), int(nBits)).astype(np.float64)


def QuantizeUniform(aNum, nBits):
    """quantize.py:12-38: the scalar form."""
    return int(_default_handle().quantize_uniform([float(aNum)], int(nBits))[0])


def Bark(f):
    """psychoac.py:27-29 (imported by codecThem.py:20)."""
    z = _default_handle().bark(np.asarray(f, dtype=np.float64))
    return z if np.ndim(f) else float(z[0])


def MSSwitchSFBands(mdct_left, mdct_right, sfBands):
    """ms_stereo.py:5-27."""
    return [int(v) for v in _default_handle().ms_switch(mdct_left, mdct_right, np.asarray(sfBands.nLines))[0]]


def StereoMaskingFactor(midThresh, sideThresh, sfBands, zVec):
    """ms_stereo.py:53-67 -> [final_midThresh, final_sideThresh] (the encoder never uses the result: psychoac.py:205-210)."""
    om, os_ = _default_handle().stereo_masking_factor(midThresh, sideThresh, zVec)
    return [om, os_]


def OverallSMRs(SMR_l, SMR_r, SMR_m, SMR_s, sfBands, ms_switch):
    """ms_stereo.py:70-81 (a per-band select; host-side)."""
    first = [SMR_m[i] if ms_switch[i] == 1 else SMR_l[i] for i in range(sfBands.nBands)]
    second = [SMR_s[i] if ms_switch[i] == 1 else SMR_r[i] for i in range(sfBands.nBands)]
    return (first, second)
