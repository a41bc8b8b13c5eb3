"""
oracle.mdct -- forward MDCT (TEST ORACLE).

Restates mdct.py:13-33 (MDCTslow forward, the O(N^2) definition) and mdct.py:63-76 (MDCT forward
via pre-twiddle -> N-point complex FFT -> post-twiddle).  Pinned by the reference's own relations
(mdct.py:185-199: MDCT == MDCTslow to 6 decimals on x = 0..1023, a = b = 512; mdct.py:131-182: TDAC
known-answer vector) in tests/test_oracle.py.  The inverse transform is decode-side, out of scope;
a minimal slow inverse is kept only for the TDAC known-answer test.
"""
import numpy as np


def MDCTslow(data, a, b, isInverse=False):
    """X[k] = (2/N) sum_n x[n] cos(2pi/N (n+n0)(k+1/2)), n0 = (b+1)/2, N = a+b  (mdct.py:21-33)."""
    N = a + b
    half = N // 2                                   # py2: N/2 integer division (mdct.py:24,26)
    n0 = (b + 1.0) / 2.0
    n = np.arange(N)
    if not isInverse:
        X = np.zeros(half)
        for k in range(half):
            X[k] = (2.0 / N) * np.dot(data, np.cos((2.0 * np.pi / N) * (np.add(n, n0)) * (k + 1.0 / 2.0)))
        return X
    x = np.zeros(N)                                 # mdct.py:42-48 (only for the TDAC test)
    k = np.arange(half)
    for i in range(N):
        x[i] = np.sum(2.0 * np.asarray(data) * np.cos((2.0 * np.pi / N) * (i + n0) * (k + 1.0 / 2.0)))
    return x


def MDCT(data, a, b, isInverse=False):
    """mdct.py:63-76: (2/N) Re( e^{-j 2pi n0 (k+1/2)/N} * FFT_N( x[n] e^{-j pi n/N} )[k] ), k < N/2."""
    if isInverse:
        raise NotImplementedError("decode side is out of scope for the encode-path oracle")
    N = a + b
    n = np.arange(N)
    n0 = (b + 1.0) / 2.0
    pre = np.exp(np.multiply(n, -1j * np.pi / N))
    spec = np.fft.fft(np.multiply(pre, data), N)
    k = np.add(np.arange(N // 2), 1.0 / 2.0)        # py2: range(0, N/2)
    post = np.exp(np.multiply(k, -1j * 2.0 * np.pi * n0 / N))
    return (2.0 / N) * np.real(np.multiply(post, spec[0:N // 2]))
