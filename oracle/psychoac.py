"""
oracle.psychoac -- psychoacoustic model: masked threshold and signal-to-mask ratios (TEST ORACLE).

Restates psychoac.py:8-29 (SPL, Intensity, Thresh, Bark), 31-78 (Masker), 82-105 (cbFreqLimits,
AssignMDCTLinesFromFreqLimits), 107-131 (ScaleFactorBands), 134-173 (getMaskedThreshold),
176-219 (CalcSMRs).  Pinned by tests/golden/ref_psychoac.npz (primitives, maskers incl. |dz| == 0.5, band tables)
and ref_smr.npz (getMaskedThreshold / CalcSMRs on every block shape at 48 and 44.1 kHz): outputs of the reference's
own functions executed through tests/golden/py2harness.py -- equal bit for bit, floats included
(tests/test_reference_golden.py).  Python-2 semantics that change results are marked `py2:`.
"""
import numpy as np
from .window import HanningWindow


def py2div(a, b):
    """Python-2 `/`: floor division when both operands are integers, true division otherwise."""
    if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)):
        return a // b
    return a / b


def SPL(intensity):
    """psychoac.py:8-12: max(96 + 10 log10(I), -30)."""
    with np.errstate(divide="ignore"):
        return np.maximum(96 + 10 * np.log10(intensity), -30,)


def Intensity(spl):
    """psychoac.py:14-18: 10^((spl-96)/10)."""
    return 10 ** ((spl - 96) / 10)


def Thresh(f):
    """psychoac.py:20-25: threshold in quiet (dB SPL) at f Hz."""
    return (3.64 * ((f / 1000.) ** (-0.8))) \
        - (6.5 * np.exp((-0.6 * (((f / 1000.) - 3.3) ** 2)))) \
        + ((10 ** (-3)) * ((f / 1000.) ** 4))


def Bark(f):
    """psychoac.py:27-29: 13 atan(0.76 f/1000) + 3.5 atan((f/7500)^2)."""
    return 13 * np.arctan(0.76 * f / 1000.) + 3.5 * np.arctan((f / 7500.) ** 2)


class Masker:
    """psychoac.py:31-78: tonal masker, 15 dB down, flat within +-0.5 Bark, -27 dB/Bark below,
    (-27 + 0.37 max(SPL-40,0)) dB/Bark above."""

    def __init__(self, f, SPL, isTonal=True):
        self.drop = 14.5 + 0.5 if isTonal else 5.5
        self.z = Bark(f)
        self.SPL = SPL
        self.f = f

    def vIntensityAtBark(self, zVec):
        dz = zVec - self.z
        above = dz > 0.5
        outside = np.abs(dz) > 0.5
        return Intensity(self.SPL - self.drop
                         + -27 * (np.abs(dz) - 0.5) * outside
                         + 0.37 * np.maximum(self.SPL - 40, 0) * (np.abs(dz) - 0.5) * outside * above)


# psychoac.py:82-84 -- the 25 Zwicker critical-band upper edges (Hz)
cbFreqLimits = [100, 200, 300, 400, 510, 630, 770, 920, 1080,
                1270, 1480, 1720, 2000, 2320, 2700, 3150, 3700,
                4400, 5300, 6400, 7700, 9500, 12000, 15500, 24000]
# pacfileThem.py:643 -- the 9 band edges used for every block that is not long+long
shortFreqLimits = [300, 630, 1080, 1720, 2700, 4400, 7700, 15500, 24000]


def AssignMDCTLinesFromFreqLimits(nMDCTLines, sampleRate, flimit=cbFreqLimits):
    """psychoac.py:86-105: band i takes the not-yet-assigned lines with centre (n+1/2) fs/(2L) < flimit[i];
    the last band takes whatever is left."""
    nMDCTLines = int(nMDCTLines)                    # py2: callers pass (a+b)/2 as an int
    centre = (np.arange(nMDCTLines) + 0.5) * ((float(sampleRate) / nMDCTLines) / 2.)
    counts = np.zeros(len(flimit))
    j = 0
    for i in range(len(flimit) - 1):
        while centre[j] < flimit[i] and j < len(centre):
            counts[i] += 1
            j += 1
    counts[len(flimit) - 1] = nMDCTLines - sum(counts)
    return counts


class ScaleFactorBands:
    """psychoac.py:107-131: lowerLine / upperLine / nLines (int arrays) from per-band line counts."""

    def __init__(self, nLines):
        self.nBands = len(nLines)
        self.lowerLine = np.cumsum(np.append([0], nLines[0:self.nBands - 1]), dtype=int)
        self.upperLine = np.cumsum(np.transpose(nLines), dtype=int) - 1
        self.nLines = self.upperLine - self.lowerLine + 1


def getMaskedThreshold(data, MDCTdata, MDCTscale, sampleRate, sfBands):
    """psychoac.py:134-173.  MDCTdata is only used for its length; MDCTscale and sfBands are unused."""
    nl = len(MDCTdata)
    n = np.arange(nl)
    MDCTFreq = (n + 0.5) * ((float(sampleRate) / nl) / 2.)
    N = len(data)
    X = np.fft.fft(HanningWindow(data))
    XI = 4. * (np.abs(X) ** 2.) / ((N ** 2.) * (3. / 8.))
    totalMask = Intensity(Thresh(MDCTFreq))
    binHz = py2div(sampleRate, N)                   # py2: 48000/2048 == 23 (psychoac.py:165)
    XI0 = XI[0]
    XI1 = XI[1]
    for i in range(2, N // 2 - 100):                # py2: range(2, N/2-100) (psychoac.py:160)
        XI2 = XI[i]
        if XI1 > XI0 and XI1 > XI2:
            level = SPL(XI0 + XI1 + XI2)
            f = binHz * (n[i - 2] * XI0 + n[i - 1] * XI1 + n[i] * XI2) / (XI0 + XI1 + XI2)
            totalMask += Masker(f, level).vIntensityAtBark(Bark(MDCTFreq))
        XI0 = XI1
        XI1 = XI2
    return SPL(totalMask)


def CalcSMRs(data, MDCTdata, MDCTscale, sampleRate, sfBands, ms=0, preCalcThresh=0.0):
    """psychoac.py:176-219.  The threshold is ALWAYS recomputed at line 210, so `ms` and
    `preCalcThresh` cannot influence the result; with ms == 0 it is evaluated twice (208, 210)."""
    SMR = np.zeros(sfBands.nBands)
    if ms != 1:
        getMaskedThreshold(data, MDCTdata, MDCTscale, sampleRate, sfBands)      # psychoac.py:208 (discarded)
    maskThresh = getMaskedThreshold(data, MDCTdata, MDCTscale, sampleRate, sfBands)
    MDCTSPL = SPL(2. * (np.abs(MDCTdata) ** 2.) / (1. / 2.)) - 6. * MDCTscale
    excess = MDCTSPL - maskThresh
    for i in range(sfBands.nBands):
        SMR[i] = np.amax(excess[sfBands.lowerLine[i]:sfBands.upperLine[i] + 1])
    return SMR
