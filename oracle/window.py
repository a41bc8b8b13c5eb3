"""
oracle.window -- analysis windows (TEST ORACLE).

Restates window.py:28-45 (HanningWindow), 49-101 (KBDWindow, alpha=4), 104-121
(TransitionWindow).  HanningWindow is pinned by tests/golden/window.npz (vectors produced by
importing the reference's window.py).  KBDWindow / TransitionWindow raise under plain NumPy 2 in the
reference (float `num` to np.linspace, window.py:60); they are pinned by tests/golden/ref_window.npz, produced by
the reference's own functions executed through tests/golden/py2harness.py (float sizes taken as NumPy < 1.12
did) -- bit-exact (tests/test_reference_golden.py) -- and additionally held to the Princen-Bradley property
win^2[n] + win^2[n+N/2] == 1 in tests/test_oracle.py.
"""
import numpy as np


def HanningWindow(dataSampleArray):
    """window.py:28-45: x[n] * (0.5 - 0.5 cos(2 pi (n+1/2)/N))."""
    N = np.size(dataSampleArray)
    n = np.add(np.linspace(0, N - 1, N), 0.5)
    w = np.add(0.5, np.multiply(-0.5, np.cos(np.multiply((2.0 * np.pi) / N, n))))
    return np.multiply(dataSampleArray, w)


def kbd_table(N, alpha=4.):
    """
    The length-N normalised Kaiser-Bessel-derived window exactly as window.py:57-98 builds it:
    kernel w[j] = I0(pi a sqrt(1-((j-M/2)/(M/2))^2))/I0(pi a), j = 0..M, M = N/2;
    rising half  win[n]      = sqrt(sum_{j<=n} w^2[j] / sum_{j=0..M} w^2[j]),
    falling half win[N/2+i]  = sqrt(sum_{j>=i+1} w^2[j] / same total),
    with the running sums taken as dense triangular-ones matrix x vector products (window.py:82-95).
    """
    half = N // 2                      # py2: N/2.0 used as an array size (window.py:58,79)
    M = N / 2.0
    j = np.linspace(0, M, half + 1)
    kernel = np.divide(np.i0(np.multiply(np.pi * alpha,
                                         np.sqrt(np.subtract(1.0, np.square(np.divide(np.subtract(j, M / 2.0), M / 2.0)))))),
                       np.i0(np.pi * alpha))
    w2 = np.square(kernel)
    total = np.sum(w2)
    ones = np.ones((half, half))
    rising = np.sqrt(np.divide(np.dot(np.tril(ones), w2[0:half]), total))
    falling = np.sqrt(np.divide(np.dot(np.triu(ones), w2[1:half + 1]), total))
    return np.concatenate((rising, falling))


def KBDWindow(dataSampleArray, alpha=4.):
    """window.py:49-101.  Rebuilds the table on every call, like the reference."""
    return np.multiply(dataSampleArray, kbd_table(np.size(dataSampleArray), alpha))


def TransitionWindow(dataSampleArray, a, b):
    """window.py:104-121: rising half of KBD(2a) on the first a samples, falling half of KBD(2b) on the last b."""
    x = np.asarray(dataSampleArray)
    left = KBDWindow(np.append(x[:a], np.zeros(a)))
    right = KBDWindow(np.append(np.zeros(b), x[a:]))
    return np.append(left[:a], right[b:])
