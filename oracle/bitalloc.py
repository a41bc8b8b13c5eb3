"""
oracle.bitalloc -- greedy water-filling bit allocation (TEST ORACLE).

Restates bitalloc.py:106-155 (BitAlloc).  Pinned bit-exactly by tests/golden/bitalloc.npz
(vectors produced by importing the reference's bitalloc.py).  The unused homework variants
(bitalloc.py:4-103) are out of scope.
"""
import numpy as np

_DEAD = -99999999999999999.0      # bitalloc.py:151 (== -1e17 exactly in binary64)


def BitAlloc(bitBudget, maxMantBits, nBands, nLines, SMR):
    """
    Repeatedly serve the band with the largest running SMR (first maximum wins ties):
    first grant = 2 bits / -12 dB, later grants = 1 bit / -6 dB; a band that is full or whose
    nLines exceed what is left is retired (value -> -1e17).  Only `nLines <= bitsLeft` is tested
    even for the 2-bit grant, so the remainder can go negative (bitalloc.py:142-146).
    `SMR` is updated IN PLACE when it is an ndarray, as in the reference (bitalloc.py:132).
    Returns (bits as float64[nBands], int(bitsLeft) truncated toward zero).
    """
    running = SMR
    left = bitBudget
    retired = 0
    bits = np.zeros(nBands)
    while left > 0:
        i = np.argmax(running)
        if bits[i] < maxMantBits and nLines[i] <= left:
            if bits[i] == 0:
                bits[i] += 2
                left -= 2 * nLines[i]
                running[i] -= 12.0
            else:
                bits[i] += 1
                left -= nLines[i]
                running[i] -= 6.0
        else:
            running[i] = _DEAD
            retired += 1
            if retired == nBands:
                break
    return (bits, int(left))
