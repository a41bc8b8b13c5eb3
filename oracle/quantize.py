"""
oracle.quantize -- mid-tread uniform quantiser and block-floating-point codes (TEST ORACLE).

Restates quantize.py:12-38 (QuantizeUniform), 61-87 (vQuantizeUniform), 90-111
(vDequantizeUniform, the PCM->float map used by pcmfile.py:98), 114-146 (ScaleFactor),
222-249 (Mantissa), 294-322 (vMantissa) of the reference.  Pinned bit-exactly by
tests/golden/quantize.npz (vectors produced by importing the reference's quantize.py).
"""
import math
import numpy as np


def _code_width(nScaleBits, nMantBits):
    # quantize.py:120,300 -- R = 2^Rs - 1 + Rm bits for the underlying uniform code
    return int((1 << int(nScaleBits)) - 1 + int(nMantBits))


def QuantizeUniform(aNum, nBits):
    """quantize.py:12-38: sign-magnitude code; |x|>=1 clips to 2^(R-1)-1; else trunc(((2^R-1)|x|+1)/2)."""
    nBits = int(nBits)
    negative = 0 if aNum >= 0.0 else 1
    mag = abs(aNum)
    if mag >= 1:
        code = (1 << (nBits - 1)) - 1
    else:
        code = int(np.int64(((float((1 << nBits) - 1)) * mag + 1) / 2))
    return (negative << (nBits - 1)) + code


def vQuantizeUniform(aNumVec, nBits):
    """quantize.py:61-87, returns float64 like the reference (sign term is a float array)."""
    nBits = int(nBits)
    x = np.asarray(aNumVec, dtype=np.float64)
    top = float(1 << (nBits - 1))
    mag = np.abs(x)
    inside = mag < 1.0
    # reference order: (inside*(2^R-1))*mag + 1, /2, *inside ; clipped entries then forced to 2^(R-1)-1
    t = ((inside * (2.0 ** nBits - 1.0)) * mag + 1.0) / 2.0 * inside
    t = np.where(t == 0.0, top - 1.0, t)
    code = t.astype(np.int64)                 # truncation toward zero (values are >= 0)
    code = np.where(mag == 0.0, 0, code)      # exact zeros stay zero
    return np.where(x < 0.0, top, 0.0) + code


def vDequantizeUniform(aQuantizedNumVec, nBits):
    """quantize.py:90-111: x = sign * 2*|code| / (2^R - 1).  (pcmfile.py:98 uses it with R=16.)"""
    q = np.asarray(aQuantizedNumVec, dtype=np.float64)
    top = 2.0 ** (nBits - 1)
    neg = q >= top
    mag = np.where(neg, q - top, q)
    sgn = np.where(neg, -1.0, 1.0)
    return (sgn * mag) * 2.0 / (2.0 ** nBits - 1.0)


def ScaleFactor(aNum, nScaleBits=3, nMantBits=5):
    """quantize.py:114-146: number of leading zeros of the (R-1)-bit magnitude code, capped at 2^Rs-1."""
    nBits = _code_width(nScaleBits, nMantBits)
    quant = QuantizeUniform(aNum, nBits)
    half = 1 << (nBits - 1)
    magCode = quant - half if quant >= half else quant
    # quantize.py:134-137: int(math.log(m, 2)); kept as-is (libm quotient), 0 for m == 0
    top_bit = 0 if magCode == 0 else int(math.log(magCode, 2))
    lz = (nBits - 2) - top_bit
    cap = (1 << int(nScaleBits)) - 1
    return lz if lz < cap else cap


def Mantissa(aNum, scale, nScaleBits=3, nMantBits=5):
    """quantize.py:222-249 (scalar block-FP mantissa; the encoder uses the vector form)."""
    nBits = _code_width(nScaleBits, nMantBits)
    quant = QuantizeUniform(aNum, nBits)
    half = 1 << (nBits - 1)
    if quant >= half:
        mag, sgn = quant - half, 1 << (int(nMantBits) - 1)
    else:
        mag, sgn = quant, 0
    cap = (1 << int(nScaleBits)) - 1
    return sgn + (mag if scale == cap else mag >> (cap - int(scale)))


def vMantissa(aNumVec, scale, nScaleBits=3, nMantBits=5):
    """quantize.py:294-322: per-line block-FP mantissas, float64 integer-valued like the reference."""
    nBits = _code_width(nScaleBits, nMantBits)
    q = vQuantizeUniform(aNumVec, nBits)
    half = float(1 << (nBits - 1))
    neg = q >= half
    mag = np.where(neg, q - half, q)
    sgn = np.where(neg, float(1 << (int(nMantBits) - 1)), 0.0)
    cap = (1 << int(nScaleBits)) - 1
    if scale == cap:
        return sgn + mag
    return sgn + np.right_shift(mag.astype(np.uint64), np.uint64(cap - int(scale)))
