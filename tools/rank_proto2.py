#!/usr/bin/env python3
"""
Numerical study (development aid) for smr_kernel's slope-node form of the upper-side spreading sum:

  U_k = sum_{m < nUp_k} I_m 2^{s_m (zq_k - z_m)},  zq_k = z_k - 1/2,  maskers and lines sorted in Bark

with EQUISPACED slope nodes sigma_r = sigma_0 - r h (r = 0 .. R-1, sigma_0 the shallowest), Lagrange weights
lambda_r(s_m) in product form, per-masker terms G[m, r] = I_m lambda_r 2^{-sigma_0 z_m} (2^{h z_m})^r, prefix sums Q_r over
maskers, and per line E0 = 2^{sigma_0 zq}, g = 2^{-h zq}, Horner in g over Q_r[nUp_k]: two 2^x per line and per masker.
Everything in float64 as the kernel would do it; the reference value is a long-double direct sum.
Reports the error of the total masked intensity in units of 2^-53 of that total, per corpus.
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from mrcaudiocodec_amd import synth
from oracle import fast, psychoac as ps

LOG2_10 = np.log2(10.0)
LD = np.longdouble


def maskers(block, N=2048, fs=48000):
    X = np.fft.fft(block * fast._hann(N))
    xi = 4. * (np.abs(X) ** 2.) / ((N ** 2.) * (3. / 8.))
    last = N // 2 - 100
    c = xi[1:last - 1]
    pk = np.nonzero((c > xi[0:last - 2]) & (c > xi[2:last]))[0] + 1
    s3 = xi[pk - 1] + xi[pk] + xi[pk + 1]
    lvl = ps.SPL(s3)
    f = (fs // N) * ((pk - 1) * xi[pk - 1] + pk * xi[pk] + (pk + 1) * xi[pk + 1]) / s3
    return 10 ** ((lvl - 15 - 96) / 10), ps.Bark(f), ((-27 + 0.37 * np.maximum(lvl - 40, 0)) / 10) * LOG2_10, lvl


def lagrange_int_nodes(theta, R):
    """lambda_r(theta) for nodes 0..R-1 in product form (prefix / suffix products), float64. theta [P] -> [P, R]"""
    P = len(theta)
    pre = np.ones((P, R))
    suf = np.ones((P, R))
    for r in range(1, R):
        pre[:, r] = pre[:, r - 1] * (theta - (r - 1))
    for r in range(R - 2, -1, -1):
        suf[:, r] = suf[:, r + 1] * (theta - (r + 1))
    from math import factorial
    c = np.array([(-1.0) ** (R - 1 - r) / (factorial(r) * factorial(R - 1 - r)) for r in range(R)])
    return pre * suf * c[None, :]


def ex2(a, x):
    """2^(a x) with the product exact (the kernel's exp2_tab64 takes the remainder by fma) and ~1 ulp of result"""
    return np.exp2(LD(a) * np.asarray(x).astype(LD)).astype(np.float64)


def upper_exact(I, z, s, zq):
    d = zq[:, None].astype(LD) - z[None, :].astype(LD)
    t = np.where(d > 0, I[None, :].astype(LD) * np.exp2(s[None, :].astype(LD) * np.maximum(d, 0)), LD(0))
    return t.sum(axis=1)


def upper_nodes(I, z, s, zq, nUp, R, margin=0.0, loc=0, hmin=1e-3):
    """equispaced nodes over [s.min, s.max] widened by `margin` node spacings on either side.
    loc > 0: every masker uses only the `loc` nodes nearest to its slope (centred stencil)."""
    a, b = s.min(), s.max()
    h = max((b - a) / (R - 1 - 2 * margin), hmin)
    sig0 = b + margin * h
    theta = (sig0 - s) / h
    if loc and loc < R:
        j0 = np.clip(np.floor(theta - (loc - 1) / 2.0 + 0.5).astype(int), 0, R - loc)
        lam = np.zeros((len(s), R))
        ll = lagrange_int_nodes(theta - j0, loc)
        for i in range(loc):
            lam[np.arange(len(s)), j0 + i] = ll[:, i]
    else:
        lam = lagrange_int_nodes(theta, R)
    F0 = I * ex2(-sig0, z)
    gm = ex2(h, z)
    G = np.empty((len(s), R))
    f = F0.copy()
    for r in range(R):
        G[:, r] = lam[:, r] * f
        f = f * gm
    Q = np.vstack([np.zeros((1, R)), np.cumsum(G, axis=0)])
    E0 = ex2(sig0, zq)
    g = ex2(-h, zq)
    acc = Q[nUp, R - 1]
    for r in range(R - 2, -1, -1):
        acc = acc * g + Q[nUp, r]
    return acc * E0, h, np.abs(lam).sum(axis=1).max()


def study(name, blocks, Rs, fs=48000, halfN=1024):
    fr = (np.arange(halfN) + 0.5) * ((float(fs) / halfN) / 2.)
    zb = ps.Bark(fr)
    quiet = ps.Intensity(ps.Thresh(fr))
    zq = zb - 0.5
    worst = {}
    info = []
    for blk in blocks:
        I, z, s, lvl = maskers(blk, fs=fs)
        if len(I) < 2:
            continue
        nUp = np.searchsorted(z, zq, side="left")
        nUp = np.array([np.count_nonzero(zb[k] - z > 0.5) for k in range(halfN)])
        cnt = np.array([np.count_nonzero(zb[k] - z >= -0.5) for k in range(halfN)])
        csum = np.concatenate([[0.0], np.cumsum(I)])
        inband = csum[cnt] - csum[nUp]
        low = np.array([(I[cnt[k]:] * np.exp2(-2.7 * LOG2_10 * (z[cnt[k]:] - zb[k] - 0.5))).sum() for k in range(halfN)])
        Uex = upper_exact(I, z, s, zq)
        base = quiet + inband + low
        total = (base.astype(LD) + Uex)
        info.append((len(I), s.min(), s.max()))
        for key in Rs:
            R, margin, loc = key
            U, h, leb = upper_nodes(I, z, s, zq, nUp, R, margin, loc)
            err = np.abs(U.astype(LD) - Uex) / total
            e = float(err.max()) / 2.0 ** -53
            w = worst.setdefault(key, [0.0, 0.0, 0.0])
            w[0] = max(w[0], e); w[1] = max(w[1], h); w[2] = max(w[2], leb)
    P = [i[0] for i in info]
    print("%-14s frames %3d  P %3d..%3d  slope %.2f..%.2f  widest %.2f" % (
        name, len(info), min(P), max(P), min(i[1] for i in info), max(i[2] for i in info), max(i[2] - i[1] for i in info)))
    for key in Rs:
        w = worst[key]
        print("     R=%2d margin=%.1f loc=%2d : max err %9.1f eps   h<=%.3f  sum|lambda|<=%.1f" % (key + tuple(w)))


def frames_of(x, n, start=1):
    return [x[(start + i) * 1024:(start + i + 2) * 1024] for i in range(n)]


if __name__ == "__main__":
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    Rs = [(12, 0.0, 0), (16, 0.0, 0), (16, 1.0, 0), (20, 0.0, 0), (20, 2.0, 0), (24, 0.0, 16), (24, 0.0, 12), (32, 0, 16)]
    study("noise", frames_of(synth.c2_noise(nfr + 2), nfr), Rs)
    study("quiet-noise", frames_of(synth.c2_noise(nfr + 2, sigma=0.001), nfr), Rs)
    study("loud-noise", frames_of(synth.c2_noise(nfr + 2, sigma=0.3), nfr), Rs)
    xs = synth.c3_stereo(nfr + 2)
    study("c3-M", frames_of(0.5 * (xs[0] + xs[1]), nfr), Rs)
    study("c3-S", frames_of(0.5 * (xs[0] - xs[1]), nfr), Rs)
    # loud band-limited noise: a cliff in the spectrum
    rng = np.random.default_rng(5)
    g = rng.normal(0, 1, (nfr + 3) * 1024)
    Gf = np.fft.rfft(g); Gf[int(len(Gf) * 4000 / 24000):] = 0; g = np.fft.irfft(Gf)
    g = synth.pcm_to_float(np.clip(np.rint(g / g.std() * 0.25 * 32767), -32767, 32767))
    study("cliff-4k", frames_of(g, nfr), Rs)
    xt, _ = synth.c4_transients(nfr + 3)
    study("transient", frames_of(xt, nfr), Rs)
    study("varied", frames_of(synth.c6_varied(4 * nfr + 2), 4 * nfr), Rs)


# ---- the a-posteriori check of the kernel: bound_k = psi* W[nUp_k] + K eps E0_k V[nUp_k] <= tol t_k
def check_study(name, blocks, R=16, margin=1.0, K=100.0, tol=2e-12, hmax=0.22, fs=48000, halfN=1024):
    from math import factorial
    fr = (np.arange(halfN) + 0.5) * ((float(fs) / halfN) / 2.)
    zb = ps.Bark(fr); quiet = ps.Intensity(ps.Thresh(fr)); zq = zb - 0.5
    nl = 0; nfail = 0; worst_pass = 0.0; worst_ratio = 0.0; nframes = 0; nq = 0; worst_fail_err = 0.0
    for blk in blocks:
        I, z, s, lvl = maskers(blk, fs=fs)
        nframes += 1
        if len(I) < 48 or len(I) > 325:
            continue
        a, b = s.min(), s.max()
        h = max((b - a) / (R - 1 - 2 * margin), 1e-3)
        if h > hmax:
            continue
        nq += 1
        nUp = np.array([np.count_nonzero(zb[k] - z > 0.5) for k in range(halfN)])
        cnt = np.array([np.count_nonzero(zb[k] - z >= -0.5) for k in range(halfN)])
        csum = np.concatenate([[0.0], np.cumsum(I)])
        inband = csum[cnt] - csum[nUp]
        low = np.array([(I[cnt[k]:] * np.exp2(-2.7 * LOG2_10 * (z[cnt[k]:] - zb[k] - 0.5))).sum() for k in range(halfN)])
        Uex = upper_exact(I, z, s, zq)
        total = (quiet + inband + low).astype(LD) + Uex
        U, h, leb = upper_nodes(I, z, s, zq, nUp, R, margin, 0)
        sig0 = b + margin * h
        theta = (sig0 - s) / h
        lam = lagrange_int_nodes(theta, R)
        A = np.abs(np.prod(theta[:, None] - np.arange(R)[None, :], axis=1)) / factorial(R)
        Lm = np.abs(lam).sum(axis=1)
        F0 = I * ex2(-sig0, z)
        W = np.concatenate([[0.0], np.cumsum(I * A)])
        V = np.concatenate([[0.0], np.cumsum(Lm * F0)])
        psi = (h * R / abs(sig0)) ** R * np.exp(-R)
        E0 = ex2(sig0, zq)
        t = (quiet + inband + low) + U
        bound = psi * W[nUp] + K * 2.0 ** -53 * E0 * V[nUp]
        ok = bound <= tol * t
        err = (np.abs(U.astype(LD) - Uex) / total).astype(np.float64)
        nl += halfN; nfail += int((~ok).sum())
        if ok.any():
            worst_pass = max(worst_pass, err[ok].max())
            worst_ratio = max(worst_ratio, (err[ok] / (bound[ok] / t[ok])).max())
        if (~ok).any():
            worst_fail_err = max(worst_fail_err, err[~ok].max())
    print("%-12s frames %3d qualify %3d | lines %6d failed check %5d (%.2f %%) | worst err among passed %.1f eps, err/bound <= %.3f | worst err among failed %.3g eps"
          % (name, nframes, nq, nl, nfail, 100.0 * nfail / max(nl, 1), worst_pass / 2.0 ** -53, worst_ratio, worst_fail_err / 2.0 ** -53))


def corpora(nfr):
    out = [("noise", frames_of(synth.c2_noise(nfr + 2), nfr)),
           ("quiet-noise", frames_of(synth.c2_noise(nfr + 2, sigma=0.001), nfr)),
           ("loud-noise", frames_of(synth.c2_noise(nfr + 2, sigma=0.3), nfr))]
    xs = synth.c3_stereo(nfr + 2)
    out += [("c3-L", frames_of(xs[0], nfr)), ("c3-R", frames_of(xs[1], nfr)),
            ("c3-M", frames_of(0.5 * (xs[0] + xs[1]), nfr)), ("c3-S", frames_of(0.5 * (xs[0] - xs[1]), nfr))]
    rng = np.random.default_rng(5)
    for cut, amp in ((4000, 0.25), (1500, 0.2), (9000, 0.05)):
        g = rng.normal(0, 1, (nfr + 3) * 1024)
        Gf = np.fft.rfft(g); Gf[int(len(Gf) * cut / 24000):] = 0; g = np.fft.irfft(Gf)
        g = synth.pcm_to_float(np.clip(np.rint(g / g.std() * amp * 32767), -32767, 32767))
        out.append(("cliff-%d" % cut, frames_of(g, nfr)))
    # the same cliff without the 16-bit floor above it (float input: digital silence above the cut)
    g = rng.normal(0, 1, (nfr + 3) * 1024)
    Gf = np.fft.rfft(g); Gf[int(len(Gf) * 4000 / 24000):] = 0; g = np.fft.irfft(Gf)
    out.append(("cliff-f64", frames_of(g / g.std() * 0.05, nfr)))
    xt, _ = synth.c4_transients(nfr + 3)
    out.append(("transient", frames_of(xt, nfr)))
    out.append(("varied", frames_of(synth.c6_varied(4 * nfr + 2), 4 * nfr)))
    return out


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "check":
    for name, blocks in corpora(nfr):
        check_study(name, blocks)
