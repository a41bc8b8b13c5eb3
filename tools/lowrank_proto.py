#!/usr/bin/env python3
"""
Numerical feasibility study (development aid): upper-slope spreading as a rank-R expansion in the slope.
  sum_{m below k} I_m 2^{s_m (z_k - z_m - 1/2)}  ~=  sum_r 2^{sigma_r z_k} * prefix_r[nUp_k],
  prefix_r[j] = sum_{m<j} I_m lambda_r(s_m) 2^{-sigma_r (z_m+1/2)},  lambda = Lagrange basis on R Chebyshev nodes
Compares the resulting masked threshold with the direct evaluation, in dB.
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from mrcaudiocodec_amd import synth
from oracle import fast, psychoac as ps

LOG2_10 = np.log2(10.0)


def maskers(block, N=2048):
    X = np.fft.fft(block * fast._hann(N))
    xi = 4. * (np.abs(X) ** 2.) / ((N ** 2.) * (3. / 8.))
    last = N // 2 - 100
    c = xi[1:last - 1]
    pk = np.nonzero((c > xi[0:last - 2]) & (c > xi[2:last]))[0] + 1
    s3 = xi[pk - 1] + xi[pk] + xi[pk + 1]
    lvl = ps.SPL(s3)
    f = 23 * ((pk - 1) * xi[pk - 1] + pk * xi[pk] + (pk + 1) * xi[pk + 1]) / s3
    return 10 ** ((lvl - 15 - 96) / 10), ps.Bark(f), ((-27 + 0.37 * np.maximum(lvl - 40, 0)) / 10) * LOG2_10


def exp2_ld(x):
    return np.exp2(x.astype(np.longdouble)).astype(np.float64) if hasattr(x, "astype") else float(np.exp2(np.longdouble(x)))


def direct_upper(I, z, s, zb):
    u = zb[:, None] - z[None, :] - 0.5
    t = np.where(u > 0, I[None, :] * np.exp2(s[None, :] * np.maximum(u, 0)), 0.0)
    return t.sum(axis=1)


def lowrank_upper(I, z, s, zb, R):
    a, b = s.min(), s.max()
    if b - a < 1e-9:
        b = a + 1e-9
    r = np.arange(R)
    sig = 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * (2 * r + 1) / (2 * R))
    w = (-1.0) ** r * np.sin(np.pi * (2 * r + 1) / (2 * R))
    d = s[:, None] - sig[None, :]
    hit = d == 0
    d = np.where(hit, 1.0, d)
    q = w[None, :] / d
    lam = q / q.sum(axis=1, keepdims=True)
    lam = np.where(hit.any(axis=1, keepdims=True), hit.astype(float), lam)
    # exponents in extended precision (stands for the kernel's double-double exponent)
    G = I[:, None] * lam * np.exp2((-sig[None, :].astype(np.longdouble)) * (z[:, None].astype(np.longdouble) + 0.5)).astype(np.float64)
    PS = np.vstack([np.zeros((1, R)), np.cumsum(G, axis=0)])
    nUp = np.searchsorted(z, zb - 0.5, side="left")          # maskers with z_m < z_k - 1/2
    E = np.exp2(sig[None, :].astype(np.longdouble) * zb[:, None].astype(np.longdouble)).astype(np.float64)
    return (E * PS[nUp]).sum(axis=1)


def study(name, block):
    halfN = 1024
    zb = ps.Bark((np.arange(halfN) + 0.5) * ((48000. / halfN) / 2.))
    quiet = ps.Intensity(ps.Thresh((np.arange(halfN) + 0.5) * ((48000. / halfN) / 2.)))
    I, z, s = maskers(block)
    inband = np.array([I[(np.abs(zb[k] - z) <= 0.5)].sum() for k in range(halfN)])
    low = np.array([(I * np.exp2(-2.7 * LOG2_10 * np.maximum(z - zb[k] - 0.5, 0)))[z - zb[k] > 0.5].sum() for k in range(halfN)])
    ref = quiet + inband + low + direct_upper(I, z, s, zb)
    out = [name, len(I), round(float(s.min()), 2), round(float(s.max()), 2)]
    for R in (8, 12, 16, 20, 24, 32):
        got = quiet + inband + low + lowrank_upper(I, z, s, zb, R)
        out.append("R%d:%.1e" % (R, np.abs(10 * np.log10(got) - 10 * np.log10(ref)).max()))
    print(*out)


if __name__ == "__main__":
    x = synth.c2_noise(6)
    study("noise", x[1024:3072])
    study("quiet-noise", synth.c2_noise(6, sigma=0.001)[1024:3072])
    study("sine", synth.c1_sine(4)[1024:3072])
    n = np.arange(2048)
    mix = synth.pcm_to_float(np.rint(12000 * np.sin(2 * np.pi * 440 * n / 48000) + 300 * np.sin(2 * np.pi * 9000 * n / 48000)
                                     + np.random.default_rng(1).normal(0, 20, 2048)))
    study("2tones+noise", mix)
    xt, _ = synth.c4_transients(10)
    study("transient", xt[4 * 1024:6 * 1024])
    study("loud-noise", synth.c2_noise(6, sigma=0.3)[1024:3072])
