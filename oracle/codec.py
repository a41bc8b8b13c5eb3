"""
oracle.codec -- per-block encode orchestration (TEST ORACLE, "faithful" flavour).

Restates codecThem.py:281-354 (EncodeSingleChannel), 359-574 (JointEncodeChannels),
205-231 (Encode), 234-260 (EncodeNoHuff), 262-278 (JointEncode), 136-203 (calculateHuffmanGain),
audiofile.py:51-53 (CodingParams) and the block framing / band-table choice of
pacfileThem.py:628-645, 799-816, 1105-1121.  Pinned by tests/golden/ref_encode.npz: the reference's own
functions executed through tests/golden/py2harness.py (see oracle/__init__.py); bit-exact.

One block at a time, with the reference's redundancy kept (window tables rebuilt per call, the
masked threshold evaluated twice per CalcSMRs, two extra thresholds for the dead M/S masking
factor): this is the code bench.py times as `cpu_baseline` (kind "port").
"""
import numpy as np

from .window import TransitionWindow
from .mdct import MDCT
from .quantize import ScaleFactor, vMantissa
from .ms_stereo import MSSwitchSFBands, StereoMaskingFactor, OverallSMRs
from .psychoac import (CalcSMRs, getMaskedThreshold, Bark, ScaleFactorBands,
                       AssignMDCTLinesFromFreqLimits, shortFreqLimits, py2div)
from .bitalloc import BitAlloc
from .huffman_tables import TABLES, TABLE_ORDER, RAW_TABLE_ID


class CodingParams:
    """audiofile.py:51-53: attribute bag shared between file layer and codec."""
    pass


def default_params(sampleRate=48000, nChannels=1, targetBitsPerSample=2.86):
    """The reference CLI defaults (pacfileThem.py:1105-1121) + the long-block band table."""
    cp = CodingParams()
    cp.sampleRate = sampleRate
    cp.nChannels = nChannels
    cp.nMDCTLines = 1024
    cp.nScaleBits = 4
    cp.nMantSizeBits = 4
    cp.targetBitsPerSample = targetBitsPerSample
    cp.nSamplesPerBlock = cp.nMDCTLines
    cp.bitReservoir = 0
    cp.nSamplesShort = 128
    cp.a = cp.nMDCTLines
    cp.b = cp.nMDCTLines
    cp.blkswBitA = 1
    cp.blkswBitB = 1
    cp.sfBands = bands_for_block(cp.a, cp.b, cp.nMDCTLines, cp.sampleRate)
    return cp


def bands_for_block(a, b, nMDCTLines, sampleRate):
    """pacfileThem.py:637-645 / 808-816: 25 critical bands for long+long, else the 9-band table."""
    half = py2div(a + b, 2)
    if a + b == 2 * nMDCTLines:
        return ScaleFactorBands(AssignMDCTLinesFromFreqLimits(half, sampleRate))
    return ScaleFactorBands(AssignMDCTLinesFromFreqLimits(half, sampleRate, shortFreqLimits))


def _max_mant_bits(cp):
    m = 1 << cp.nMantSizeBits                       # codecThem.py:292-293
    return 16 if m > 16 else m


def _quantise_stream(lines_for_band, bitAlloc, sfBands, nScaleBits, halfN):
    """codecThem.py:335-350 / 510-559: per-band scale factor + mantissas; bands with 0 bits are omitted."""
    scaleFactor = np.empty(sfBands.nBands, dtype=np.int32)
    nMant = halfN
    for iBand in range(sfBands.nBands):
        if not bitAlloc[iBand]:
            nMant -= sfBands.nLines[iBand]
    mantissa = np.empty(int(nMant), dtype=np.int32)  # py2: halfN is a float used as a size
    iMant = 0
    for iBand in range(sfBands.nBands):
        lo = sfBands.lowerLine[iBand]
        hi = sfBands.upperLine[iBand] + 1
        nLines = sfBands.nLines[iBand]
        lines = lines_for_band(iBand)
        peak = np.max(np.abs(lines[lo:hi]))
        scaleFactor[iBand] = ScaleFactor(peak, nScaleBits, bitAlloc[iBand])
        if bitAlloc[iBand]:
            mantissa[iMant:iMant + nLines] = vMantissa(lines[lo:hi], scaleFactor[iBand], nScaleBits, bitAlloc[iBand])
            iMant += nLines
    return scaleFactor, mantissa


def EncodeSingleChannel(data, codingParams):
    """codecThem.py:281-354.  Returns (scaleFactor int32[nBands], bitAlloc int[nBands],
    mantissa int32[compact], overallScale int); writes codingParams.bitReservoir (line 332)."""
    cp = codingParams
    halfN = (cp.a + cp.b) / 2.
    half = int(halfN)
    nScaleBits = cp.nScaleBits
    sfBands = cp.sfBands
    bitBudget = cp.targetBitsPerSample * halfN
    bitBudget -= nScaleBits * (sfBands.nBands + 1)
    bitBudget -= cp.nMantSizeBits * sfBands.nBands
    bitBudget -= cp.blkswBitA
    bitBudget -= cp.blkswBitB
    bitBudget += cp.bitReservoir

    mdctLines = MDCT(TransitionWindow(data, cp.a, cp.b), cp.a, cp.b)[:half]
    overallScale = ScaleFactor(np.max(np.abs(mdctLines)), nScaleBits)
    mdctLines *= (1 << overallScale)

    SMRs = CalcSMRs(data, mdctLines, overallScale, cp.sampleRate, sfBands)
    (bitAlloc, remaining) = BitAlloc(bitBudget, _max_mant_bits(cp), sfBands.nBands, sfBands.nLines, SMRs)
    bitAlloc = bitAlloc.astype(int)
    cp.bitReservoir = int(remaining)

    scaleFactor, mantissa = _quantise_stream(lambda iBand: mdctLines, bitAlloc, sfBands, nScaleBits, halfN)
    return (scaleFactor, bitAlloc, mantissa, overallScale)


def JointEncodeChannels(dataLeft, dataRight, codingParams):
    """codecThem.py:359-574.  Returns ([sf1,sf2],[ba1,ba2],[m1,m2],[osL,osR,osM,osS],ms_switch);
    stream 1 carries Mid-or-Left per band, stream 2 Side-or-Right; writes codingParams.bitReservoir."""
    cp = codingParams
    dataLeft = np.asarray(dataLeft, dtype=np.float64)
    dataRight = np.asarray(dataRight, dtype=np.float64)
    dataMid = (dataLeft + dataRight) / 2.0
    dataSide = (dataLeft - dataRight) / 2.0
    N = cp.a + cp.b
    halfN = (cp.a + cp.b) / 2.
    half = int(halfN)
    nScaleBits = cp.nScaleBits
    sfBands = cp.sfBands

    bitBudget = cp.targetBitsPerSample * halfN
    bitBudget -= nScaleBits * (sfBands.nBands)
    bitBudget -= cp.nMantSizeBits * sfBands.nBands
    bitBudget += bitBudget
    bitBudget -= sfBands.nBands
    bitBudget -= nScaleBits * 4
    bitBudget += cp.bitReservoir
    bitBudget -= cp.blkswBitA
    bitBudget -= cp.blkswBitB

    time = [dataLeft, dataRight, dataMid, dataSide]
    lines = [MDCT(TransitionWindow(x, cp.a, cp.b), cp.a, cp.b)[:half] for x in time]

    ms_switch = MSSwitchSFBands(lines[0], lines[1], sfBands)       # on the UNSCALED L/R lines (line 436)

    overall = []
    for X in lines:                                                 # order L, R, M, S (lines 440-460)
        s = ScaleFactor(np.max(np.abs(X)), nScaleBits)
        X *= (1 << s)
        overall.append(s)

    # codecThem.py:465-476 -- M/S masking-level-difference thresholds.  Their result is handed to
    # CalcSMRs, which ignores it (psychoac.py:205-210); evaluated here only to keep the cost faithful.
    midT = getMaskedThreshold(time[2], lines[2], overall[2], cp.sampleRate, sfBands)
    sideT = getMaskedThreshold(time[3], lines[3], overall[3], cp.sampleRate, sfBands)
    freq = np.multiply(np.add(np.linspace(0, N - 1, N), 0.5), py2div(cp.sampleRate, N))
    newT = StereoMaskingFactor(midT, sideT, sfBands, Bark(freq[0:N // 2]))

    smr = [CalcSMRs(time[0], lines[0], overall[0], cp.sampleRate, sfBands),
           CalcSMRs(time[1], lines[1], overall[1], cp.sampleRate, sfBands),
           CalcSMRs(time[2], lines[2], overall[2], cp.sampleRate, sfBands, 1, newT[0]),
           CalcSMRs(time[3], lines[3], overall[3], cp.sampleRate, sfBands, 1, newT[1])]
    (SMR1, SMR2) = OverallSMRs(smr[0], smr[1], smr[2], smr[3], sfBands, ms_switch)

    nLinesPass = np.append(sfBands.nLines, sfBands.nLines)
    SMRsPass = np.append(SMR1, SMR2)
    (bitAlloc, remaining) = BitAlloc(bitBudget, _max_mant_bits(cp), 2 * sfBands.nBands, nLinesPass, SMRsPass)
    bitAlloc = bitAlloc.astype(int)
    bitAlloc1 = bitAlloc[0:sfBands.nBands]
    bitAlloc2 = bitAlloc[sfBands.nBands:]
    cp.bitReservoir = int(remaining)

    sf1, m1 = _quantise_stream(lambda i: lines[2] if ms_switch[i] == 1 else lines[0],
                               bitAlloc1, sfBands, nScaleBits, halfN)
    sf2, m2 = _quantise_stream(lambda i: lines[3] if ms_switch[i] == 1 else lines[1],
                               bitAlloc2, sfBands, nScaleBits, halfN)
    return ([sf1, sf2], [bitAlloc1, bitAlloc2], [m1, m2], overall, ms_switch)


def huffman_cost(mantissa, bitAlloc, sfBands, table, escape_value, raw_bits):
    """Bit count codecThem.py:157-173 assigns to one table (band loop stops once it exceeds raw_bits)."""
    cost = 0
    iMant = 0
    esc_len = table[escape_value][1]
    for iBand in range(sfBands.nBands):
        if cost > raw_bits:
            break
        ba = bitAlloc[iBand]
        if ba:
            for _ in range(int(sfBands.nLines[iBand])):
                v = int(mantissa[iMant])
                if v not in table and v != escape_value:
                    cost += ba + esc_len
                else:
                    cost += table[v][1]      # the escape VALUE itself is priced as its code only (lines 169-172)
                iMant += 1
    return cost


def calculateHuffmanGain(mantissa, bitAlloc, codingParams):
    """codecThem.py:136-203 with the table order fixed to oracle.huffman_tables.TABLE_ORDER
    (the reference takes os.walk/glob directory order, which differs between its encoder and decoder).
    Returns (table id | 15, mantissa array | list of code strings, bits_saved)."""
    sfBands = codingParams.sfBands
    raw_bits = 0
    for iBand in range(sfBands.nBands):
        if bitAlloc[iBand]:
            raw_bits += bitAlloc[iBand] * sfBands.nLines[iBand]
    best = raw_bits
    table_to_use = RAW_TABLE_ID
    for i, name in enumerate(TABLE_ORDER):
        table, escape_value = TABLES[name]
        cost = huffman_cost(mantissa, bitAlloc, sfBands, table, escape_value, raw_bits)
        if cost < best:
            best = cost
            table_to_use = i
    if table_to_use == RAW_TABLE_ID:
        codes = mantissa
    else:
        table, escape_value = TABLES[TABLE_ORDER[table_to_use]]
        codes = []
        for v in mantissa:
            v = int(v)
            if v in table and v != escape_value:
                codes.append(table[v][0])
            else:
                codes.append(table[escape_value][0] + "/" + str(v))
    return (table_to_use, codes, raw_bits - best)


def Encode(data, codingParams):
    """codecThem.py:205-231: independent channels; channel i+1 sees channel i's reservoir + Huffman savings."""
    scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable = [], [], [], [], []
    for iCh in range(codingParams.nChannels):
        (s, b, m, o) = EncodeSingleChannel(data[iCh], codingParams)
        (table_to_use, new_m, bits_saved) = calculateHuffmanGain(m, b, codingParams)
        codingParams.bitReservoir += bits_saved
        scaleFactor.append(s); bitAlloc.append(b); mantissa.append(new_m)
        overallScaleFactor.append(o); huffTable.append(table_to_use)
    return (scaleFactor, bitAlloc, mantissa, overallScaleFactor, huffTable)


def EncodeNoHuff(data, codingParams):
    """codecThem.py:234-260: as Encode without the Huffman stage (table id 15 = raw)."""
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
