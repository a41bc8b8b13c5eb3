#!/usr/bin/env python3
"""
NumPy prototype of the index maps the HIP kernels use (development aid, not product, not oracle):
  * signed circular shift that turns the reference's n0=(b+1)/2 MDCT into the standard-phase MDCT,
  * fold to a length-N/2 DCT-IV, DCT-IV through an N/4-point complex FFT with pre/post twiddles,
  * mixed-radix Stockham autosort passes (the LDS FFT), real-input FFT through a half-size complex FFT.
Checked against the O(N^2) definition for all four block shapes.
"""
import numpy as np


def stockham(x, radices):
    n = len(x)
    a = np.array(x, dtype=complex)
    p = 1
    W = np.exp(-2j * np.pi * np.arange(n) / n)
    for R in radices:
        out = np.empty(n, dtype=complex)
        T = n // R
        for i in range(T):
            k = i % p
            j = (i // p) * (p * R) + k
            u = [a[i + r * T] * W[(k * r * (n // (p * R))) % n] for r in range(R)]
            for q in range(R):
                out[j + q * p] = sum(u[r] * W[((q * r) % R) * (n // R)] for r in range(R))
        a = out
        p *= R
    return a


def mdct_n4(xw, a, b, radices=None):
    """xw: windowed block of length N=a+b.  Returns N/2 lines = (2/N) sum xw[n] cos(2pi/N (n+(b+1)/2)(k+1/2))."""
    N = a + b
    M = N // 2
    Q = N // 4
    d = (b - a) // 4                       # n0 - (N/4 + 1/2)
    y = np.empty(N)
    for n in range(N):                      # signed circular shift: y[(n+d) mod N] = +-xw[n]
        m = n + d
        if m < 0:
            y[m + N] = -xw[n]
        elif m >= N:
            y[m - N] = -xw[n]
        else:
            y[m] = xw[n]
    u = np.empty(M)
    h = M // 2
    for n in range(h):
        u[n] = -y[3 * h - 1 - n] - y[3 * h + n]
        u[h + n] = y[n] - y[2 * h - 1 - n]
    n = np.arange(Q)
    t = (u[2 * n] + 1j * u[M - 1 - 2 * n]) * np.exp(-1j * np.pi * (4 * n + 1) / (4 * M))
    Tt = stockham(t, radices) if radices else np.fft.fft(t)
    k = np.arange(Q)
    c = Tt * np.exp(-1j * np.pi * k / M)            # = exp(-i pi (4k+1)/(4M)) * exp(+i pi/(4M)) folded below
    c = Tt * np.exp(-1j * np.pi * (4 * k) / (4 * M))
    X = np.empty(M)
    X[2 * k] = c.real
    X[M - 1 - 2 * k] = -c.imag
    return (2.0 / N) * X


def slow(xw, a, b):
    N = a + b
    n0 = (b + 1) / 2
    n = np.arange(N)
    return np.array([(2.0 / N) * np.dot(xw, np.cos(2 * np.pi / N * (n + n0) * (k + 0.5))) for k in range(N // 2)])


def rfft_half(x, radices=None):
    """Real FFT of even length N via an N/2 complex FFT; returns bins 0..N/2-1."""
    N = len(x)
    H = N // 2
    z = x[0::2] + 1j * x[1::2]
    Z = stockham(z, radices) if radices else np.fft.fft(z)
    k = np.arange(H)
    Zc = np.conj(Z[(-k) % H])
    E = 0.5 * (Z + Zc)
    O = -0.5j * (Z - Zc)
    return E + np.exp(-2j * np.pi * k / N) * O


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.normal(size=60)
    for rad in ([4, 5, 3], [2, 2, 3, 5], [3, 4, 5], [5, 3, 2, 2]):
        assert np.allclose(stockham(x, rad), np.fft.fft(x)), rad
    x = rng.normal(size=512) + 1j * rng.normal(size=512)
    assert np.allclose(stockham(x, [8, 8, 8]), np.fft.fft(x))
    for (a, b, rad) in [(1024, 1024, [4, 4, 4, 4, 2]), (128, 128, [4, 4, 4]), (1024, 128, [4, 4, 2, 3, 3]),
                        (128, 1024, [3, 3, 4, 4, 2]), (8, 8, None), (4, 4, None)]:
        xw = rng.normal(size=a + b)
        e = np.abs(mdct_n4(xw, a, b, rad) - slow(xw, a, b)).max()
        print(a, b, "mdct err", e)
        assert e < 1e-12
    for N, rad in [(2048, [4, 4, 4, 4, 4]), (256, [4, 4, 4, 2]), (1152, [4, 4, 4, 3, 3])]:
        x = rng.normal(size=N)
        e = np.abs(rfft_half(x, rad) - np.fft.fft(x)[:N // 2]).max()
        print(N, "rfft err", e)
        assert e < 1e-11
    print("ok")


def imdct_n4(X, a, b):
    """Inverse with the reference's phase (mdct.py:98-122): x[n] = sum_k 2 X[k] cos(2pi/N (n+(b+1)/2)(k+1/2)).
    DCT-IV of the lines through the same N/4-point FFT as the forward transform, then the transpose of the fold
    (unfold M -> N) and the inverse of the signed circular shift."""
    N = a + b
    M = N // 2
    Q = N // 4
    h = Q
    d = (b - a) // 4
    n = np.arange(Q)
    t = (X[2 * n] + 1j * X[M - 1 - 2 * n]) * np.exp(-1j * np.pi * (4 * n + 1) / (4 * M))
    c = np.fft.fft(t) * np.exp(-1j * np.pi * (4 * n) / (4 * M))
    v = np.empty(M)
    v[2 * n] = c.real
    v[M - 1 - 2 * n] = -c.imag                       # v = DCT-IV(X)
    y = np.empty(N)
    for i in range(h):
        y[3 * h - 1 - i] = -v[i]
        y[3 * h + i] = -v[i]
        y[i] = v[h + i]
        y[2 * h - 1 - i] = -v[h + i]
    x = np.empty(N)
    for i in range(N):
        m = i + d
        if m < 0:
            x[i] = -y[m + N]
        elif m >= N:
            x[i] = -y[m - N]
        else:
            x[i] = y[m]
    return 2.0 * x


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for (a, b) in [(1024, 1024), (128, 128), (1024, 128), (128, 1024), (4, 4)]:
        N = a + b
        X = rng.normal(size=N // 2)
        n0 = (b + 1) / 2
        k = np.arange(N // 2)
        ref = np.array([np.sum(2.0 * X * np.cos(2 * np.pi / N * (i + n0) * (k + 0.5))) for i in range(N)])
        got = imdct_n4(X, a, b)
        print("imdct", (a, b), np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
