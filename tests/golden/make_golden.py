#!/usr/bin/env python3
"""
Generate the golden vectors that pin the oracle, by IMPORTING the reference's own modules that run
under Python 3 / NumPy 2 (quantize.py, bitalloc.py, ms_stereo.py, window.py -- SURVEY.md 8c) on
synthetic inputs and recording their outputs.  Run in the build container only (the reference does
not exist on the GPU box); the resulting .npz files are data (inputs + expected outputs) and are
committed.  No reference source text is stored.

    python tests/golden/make_golden.py [/root/reference]

Not importable from the reference (py2 SyntaxError, directly or transitively): mdct, psychoac,
codecThem, pacfileThem, bitpack, huffman.  KBDWindow/TransitionWindow import but raise TypeError
(float `num` to np.linspace, window.py:60).  Those stages are pinned by relations / "parity
unpinned" as documented in oracle/__init__.py.
"""
import os
import sys
import types
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import quantize as rq          # noqa: E402
import bitalloc as rb          # noqa: E402
import ms_stereo as rm         # noqa: E402
import window as rw            # noqa: E402

rng = np.random.default_rng(20261004)


def bands(nLines):
    """Stand-in for psychoac.ScaleFactorBands (not importable): only the attributes ms_stereo reads."""
    nLines = np.asarray(nLines, dtype=int)
    o = types.SimpleNamespace()
    o.nBands = len(nLines)
    o.upperLine = np.cumsum(nLines) - 1
    o.lowerLine = o.upperLine - nLines + 1
    o.nLines = nLines
    return o


LONG = [4, 5, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 12, 14, 16, 19, 24, 30, 38, 47, 56, 76, 107, 149, 363]
SHORT = [2, 1, 3, 3, 5, 9, 18, 42, 45]
TRANS = [7, 8, 11, 15, 24, 41, 79, 187, 204]

# ------------------------------------------------------------------ quantize
vals = np.concatenate([
    [0.0, -0.0, 1.0, -1.0, 1.5, -2.0, 0.999999999, -0.999999999, 1e-9, -1e-9, 1e-300, -1e-300,
     0.5, -0.5, 0.25, 2.0 ** -15, -(2.0 ** -15), 2.0 ** -19, 2.0 ** -30, 3e-5, -3e-5],
    2.0 ** -np.arange(1, 33), -(2.0 ** -np.arange(1, 33)),
    (2.0 ** -np.arange(1, 33)) * (1 - 2.0 ** -40), (2.0 ** -np.arange(1, 33)) * (1 + 2.0 ** -40),
    rng.uniform(-1.2, 1.2, 400), rng.normal(0, 0.01, 400), rng.normal(0, 1e-4, 200)])
q = {"vals": vals}
sf_cases = [(s, m) for s in (3, 4) for m in (0, 2, 3, 4, 5, 8, 12, 16)]
q["sf_cases"] = np.array(sf_cases)
q["sf_out"] = np.array([[rq.ScaleFactor(float(v), s, m) for v in vals] for (s, m) in sf_cases], dtype=np.int64)
nbits_cases = list(range(2, 32))
q["nbits_cases"] = np.array(nbits_cases)
q["vquant_out"] = np.array([rq.vQuantizeUniform(vals, nb) for nb in nbits_cases])
q["quant_out"] = np.array([[int(rq.QuantizeUniform(float(v), nb)) for v in vals] for nb in nbits_cases], dtype=np.int64)
mant_cases = [(sc, 4, mb) for mb in (2, 3, 4, 5, 7, 11, 16) for sc in range(16)] + \
             [(sc, 3, mb) for mb in (2, 5, 9) for sc in range(8)]
q["mant_cases"] = np.array(mant_cases)
q["vmant_out"] = np.array([rq.vMantissa(vals, sc, sb, mb) for (sc, sb, mb) in mant_cases])
q["mant_out"] = np.array([[int(rq.Mantissa(float(v), sc, sb, mb)) for v in vals[:120]]
                          for (sc, sb, mb) in mant_cases], dtype=np.int64)
codes16 = np.concatenate([np.arange(0, 40), np.arange(32700, 32840), np.arange(65500, 65536),
                          rng.integers(0, 65536, 300)]).astype(np.float64)
q["codes16"] = codes16
q["vdequant16_out"] = rq.vDequantizeUniform(codes16, 16)
np.savez_compressed(os.path.join(OUT, "quantize.npz"), **q)

# ------------------------------------------------------------------ bitalloc
cases = []
def add_case(budget, maxb, nLines, smr):
    smr = np.asarray(smr, dtype=np.float64)
    nLines = np.asarray(nLines)
    work = smr.copy()
    bits, left = rb.BitAlloc(budget, maxb, len(nLines), nLines, work)
    cases.append((float(budget), int(maxb), nLines.astype(np.int64), smr, np.asarray(bits, dtype=np.float64),
                  int(left), work))

for tbl in (LONG, SHORT, TRANS):
    nb = len(tbl)
    half = sum(tbl)
    for joint in (False, True):
        nl = tbl + tbl if joint else tbl
        for _ in range(12):
            smr = rng.normal(10, 15, len(nl))
            mono_budget = 2.86 * half - 4 * (nb + 1) - 4 * nb - 2
            budget = (2 * (2.86 * half - 4 * nb - 4 * nb) - nb - 16 - 2) if joint else mono_budget
            budget += int(rng.integers(-50, 400))
            add_case(budget, 16, nl, smr)
        add_case(5, 16, nl, rng.normal(0, 5, len(nl)))            # tiny budget -> negative remainder
        add_case(1e6, 16, nl, rng.normal(0, 5, len(nl)))          # everything maxes out
        add_case(300.5, 8, nl, np.zeros(len(nl)))                 # exact ties -> lowest index
        add_case(0, 16, nl, rng.normal(0, 5, len(nl)))            # no loop
        add_case(-7.5, 16, nl, rng.normal(0, 5, len(nl)))
        add_case(2000.25, 16, nl, np.round(rng.normal(0, 6, len(nl))) * 6.0)   # many ties after -6/-12 steps
add_case(5, 16, [4, 4], [10., 5.])                                # SURVEY 8a10 probe: returns -3
b = {"n": np.array(len(cases))}
for i, (budget, maxb, nl, smr, bits, left, work) in enumerate(cases):
    b["budget_%d" % i] = np.array(budget); b["maxb_%d" % i] = np.array(maxb)
    b["nlines_%d" % i] = nl; b["smr_%d" % i] = smr
    b["bits_%d" % i] = bits; b["left_%d" % i] = np.array(left); b["smr_after_%d" % i] = work
np.savez_compressed(os.path.join(OUT, "bitalloc.npz"), **b)

# ------------------------------------------------------------------ ms_stereo
m = {}
k = 0
for tbl in (LONG, SHORT, TRANS):
    sfb = bands(tbl)
    half = sum(tbl)
    for trial in range(6):
        L = rng.normal(0, 10.0 ** rng.uniform(-5, -1), half)
        mix = rng.uniform(0, 1, half) < 0.5
        R = np.where(mix, 0.8 * L + 0.2 * rng.normal(0, 1e-3, half), rng.normal(0, 1e-3, half))
        if trial == 0:
            R = L.copy()
        if trial == 1:
            L = np.zeros(half); R = np.zeros(half)
        sw = rm.MSSwitchSFBands(L, R, sfb)
        midT = rng.uniform(-10, 60, half); sideT = rng.uniform(-10, 60, half)
        z = np.linspace(0.05, 24.5, half)
        smf = rm.StereoMaskingFactor(midT, sideT, sfb, z)
        smrs = [rng.normal(0, 20, sfb.nBands) for _ in range(4)]
        o1, o2 = rm.OverallSMRs(smrs[0], smrs[1], smrs[2], smrs[3], sfb, sw)
        m["nlines_%d" % k] = np.array(tbl); m["L_%d" % k] = L; m["R_%d" % k] = R
        m["switch_%d" % k] = np.array(sw, dtype=np.int64)
        m["midT_%d" % k] = midT; m["sideT_%d" % k] = sideT; m["z_%d" % k] = z
        m["smf0_%d" % k] = smf[0]; m["smf1_%d" % k] = smf[1]
        m["smrs_%d" % k] = np.array(smrs); m["o1_%d" % k] = np.array(o1); m["o2_%d" % k] = np.array(o2)
        k += 1
m["n"] = np.array(k)
np.savez_compressed(os.path.join(OUT, "ms_stereo.npz"), **m)

# ------------------------------------------------------------------ window (Hann + Sine only)
w = {}
for N in (2048, 1152, 256, 8):
    x = rng.normal(0, 0.3, N)
    w["x_%d" % N] = x
    w["hann_%d" % N] = rw.HanningWindow(x)
    w["sine_%d" % N] = rw.SineWindow(x)
np.savez_compressed(os.path.join(OUT, "window.npz"), **w)
print("golden vectors written to", OUT)

# ------------------------------------------------------------------ decode side ("next" row f-4)
# vDequantize / vDequantizeUniform (quantize.py:90-111, 325-357) and ReconstructLR (ms_stereo.py:33-49)
d = {}
dq_cases, dq_in, dq_out = [], [], []
for nScaleBits in (3, 4):
    cap = (1 << nScaleBits) - 1
    for nMantBits in (2, 3, 4, 5, 8, 12, 16):
        for scale in sorted({0, 1, cap // 2, cap - 1, cap}):
            mant = np.concatenate([[0, 1, (1 << (nMantBits - 1)) - 1, 1 << (nMantBits - 1), (1 << nMantBits) - 1],
                                   rng.integers(0, 1 << nMantBits, 27)]).astype(np.int32)
            dq_cases.append((scale, nScaleBits, nMantBits))
            dq_in.append(mant)
            dq_out.append(np.asarray(rq.vDequantize(scale, mant, nScaleBits, nMantBits), dtype=np.float64))
d["dq_cases"], d["dq_in"], d["dq_out"] = np.array(dq_cases), np.array(dq_in), np.array(dq_out)
du_bits = [2, 3, 8, 12, 16, 19, 24, 31]
d["du_bits"] = np.array(du_bits)
d["du_in"] = np.array([np.concatenate([[0, 1, (1 << (nb - 1)) - 1, 1 << (nb - 1), (1 << nb) - 1],
                                       rng.integers(0, 1 << nb, 27)]) for nb in du_bits], dtype=np.float64)
d["du_out"] = np.array([np.asarray(rq.vDequantizeUniform(row, nb), dtype=np.float64) for row, nb in zip(d["du_in"], du_bits)])
for name, nl in (("long", LONG), ("short", SHORT), ("trans", TRANS)):
    sfb = bands(nl)
    n = int(np.sum(nl))
    a1, a2 = rng.normal(0, 0.1, n), rng.normal(0, 0.1, n)
    sw = rng.integers(0, 2, len(nl))
    left, right = rm.ReconstructLR(a1, a2, sfb, sw)
    d["lr_%s_in1" % name], d["lr_%s_in2" % name], d["lr_%s_sw" % name] = a1, a2, sw
    d["lr_%s_left" % name], d["lr_%s_right" % name] = np.asarray(left, dtype=np.float64), np.asarray(right, dtype=np.float64)
np.savez_compressed(os.path.join(OUT, "decode.npz"), **d)
print("golden vectors written to", OUT)
