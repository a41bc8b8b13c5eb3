"""
oracle.ms_stereo -- per-band M/S decision and SMR selection (TEST ORACLE).

Restates ms_stereo.py:5-27 (MSSwitchSFBands), 53-67 (StereoMaskingFactor), 70-81 (OverallSMRs).
Pinned bit-exactly by tests/golden/ms_stereo.npz (vectors produced by importing the reference's
ms_stereo.py).  ReconstructLR (decoder side) is out of scope.
"""
import numpy as np


def MSSwitchSFBands(mdct_left, mdct_right, sfBands):
    """1 where sum|L^2-R^2| < 0.8*sum|L^2+R^2| over the band's lines (ms_stereo.py:19-22)."""
    l2 = np.square(mdct_left)
    r2 = np.square(mdct_right)
    diff = l2 - r2
    summ = l2 + r2
    out = []
    for lo, hi in zip(sfBands.lowerLine, sfBands.upperLine):
        d = np.sum(np.abs(diff[lo:hi + 1]))
        s = np.sum(np.abs(summ[lo:hi + 1]))
        out.append(1 if d < 0.8 * s else 0)
    return out


def StereoMaskingFactor(midThresh, sideThresh, sfBands, zVec):
    """ms_stereo.py:53-67.  (Its result never reaches the encoder output: psychoac.py:205-210.)"""
    zc = np.minimum(zVec, 15.5 * np.ones(np.size(zVec)))
    MLD = np.power(10.0, 1.25 * (1 - np.cos((np.pi / 15.5) * zc) - 2.5))
    mid2 = MLD * midThresh
    side2 = MLD * sideThresh
    return [np.maximum(midThresh, np.minimum(sideThresh, side2)),
            np.maximum(sideThresh, np.minimum(midThresh, mid2))]


def OverallSMRs(SMR_l, SMR_r, SMR_m, SMR_s, sfBands, ms_switch):
    """ms_stereo.py:70-81: (M,S) SMRs where the band is switched to M/S, else (L,R)."""
    first, second = [], []
    for i in range(sfBands.nBands):
        if ms_switch[i] == 1:
            first.append(SMR_m[i]); second.append(SMR_s[i])
        else:
            first.append(SMR_l[i]); second.append(SMR_r[i])
    return (first, second)
