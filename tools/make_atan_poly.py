"""Coefficients of mrc_device.hpp's atan_pos(): atan(t) = t * Q(t^2) on [0, 1], Q of degree DEG in u = t^2, from Chebyshev
interpolation of atan(sqrt(u)) / sqrt(u) carried out in 60-digit decimal arithmetic (stdlib only); prints C initialisers
and the measured error against the decimal reference."""
import sys
from decimal import Decimal as D, getcontext

getcontext().prec = 70
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 21


def pi():
    # Machin
    def atan_inv(n):
        x = D(1) / n; x2 = x * x; s = x; t = x; k = 1
        while True:
            t = -t * x2; k += 2; term = t / k
            if abs(term) < D(10) ** -68: break
            s += term
        return s
    return 4 * (4 * atan_inv(5) - atan_inv(239))


PI = pi()


def atan_dec(t):           # 0 <= t <= 1
    t = D(t)
    # argument halving twice: atan(t) = 2 atan(t / (1 + sqrt(1 + t^2)))
    n = 0
    while t > D("0.2"):
        t = t / (1 + (1 + t * t).sqrt()); n += 1
    t2 = t * t; s = t; term = t; k = 1
    while True:
        term = -term * t2; k += 2; a = term / k
        if abs(a) < D(10) ** -66: break
        s += a
    return s * (2 ** n)


def g(u):                  # atan(sqrt(u)) / sqrt(u), u in [0, 1]
    if u == 0:
        return D(1)
    r = D(u).sqrt()
    return atan_dec(r) / r


def cos_dec(x):
    x = D(x); s = D(1); term = D(1); k = 0
    while True:
        term = -term * x * x / ((k + 1) * (k + 2)); k += 2
        if abs(term) < D(10) ** -66: break
        s += term
    return s


n = DEG + 1
nodes = [cos_dec(PI * (2 * j + 1) / (2 * n)) for j in range(n)]            # Chebyshev nodes on [-1, 1]
vals = [g((x + 1) / 2) for x in nodes]
# Chebyshev coefficients
cheb = []
for k in range(n):
    s = D(0)
    for j in range(n):
        s += vals[j] * cos_dec(PI * k * (2 * j + 1) / (2 * n))
    cheb.append(s * 2 / n)
cheb[0] /= 2
# to monomials in x, then substitute x = 2u - 1
T = [[D(1)], [D(0), D(1)]]
for k in range(2, n):
    a = [D(0)] + [2 * c for c in T[k - 1]]
    b = T[k - 2] + [D(0)] * (len(a) - len(T[k - 2]))
    T.append([x - y for x, y in zip(a, b)])
px = [D(0)] * n
for k in range(n):
    for i, c in enumerate(T[k]):
        px[i] += cheb[k] * c
# p(x) with x = 2u - 1 -> q(u)
q = [D(0)] * n
pw = [D(1)]                 # (2u - 1)^i as polynomial in u
for i in range(n):
    for j, c in enumerate(pw):
        q[j] += px[i] * c
    nxt = [D(0)] * (len(pw) + 1)
    for j, c in enumerate(pw):
        nxt[j] -= c
        nxt[j + 1] += 2 * c
    pw = nxt
coef = [float(c) for c in q]
print("// atan(t) = t * Q(t^2), 0 <= t <= 1; Q of degree %d (tools/make_atan_poly.py)" % DEG)
print("constexpr double kAtanQ[%d] = {%s};" % (n, ", ".join(float.hex(c) for c in coef)))
# error check in float arithmetic (Horner as the device does it, fma emulated by exact Decimal then round)
import math
worst = 0.0
for i in range(0, 2001):
    t = i / 2000.0
    u = t * t
    p = coef[-1]
    for c in reversed(coef[:-1]):
        p = float(D(p) * D(u) + D(c))          # fma: one rounding
    val = t * p
    ref = atan_dec(t)
    if ref != 0:
        err = abs((D(val) - ref) / ref)
        worst = max(worst, float(err))
print("// max relative error of the double evaluation on [0, 1]: %.3g (2^-53 = 1.1e-16)" % worst)
print("// libm for comparison:", max(abs((D(math.atan(i / 2000.0)) - atan_dec(i / 2000.0)) / atan_dec(i / 2000.0)) for i in range(1, 2001)))
