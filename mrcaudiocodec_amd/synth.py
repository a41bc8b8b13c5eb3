"""
Synthetic 48 kHz PCM for the configurations of BASELINE.md / SURVEY.md 8(d).  int16 PCM is mapped to the
reference's signed-fraction float64 exactly as pcmfile.py:91-100 + quantize.py:90-111 do
(x = sign(c) * 2|c| / 65535), so the inputs carry a real 16-bit noise floor.  NumPy only (host side).
Every stream starts with one hop of zeros: the reference's priorBlock at file start (pacfileThem.py:615-618).
"""
import numpy as np

HOP = 1024


def pcm_to_float(pcm):
    """int16 code c -> sign(c) * 2|c| / 65535; -32768 -> 0.0 (its magnitude 2^15 is read as a bare sign bit:
    pcmfile.py:91-100 + quantize.py:90-111; pinned by tests/golden/ref_encode.npz `pcmmap_*`)."""
    p = np.asarray(pcm, dtype=np.float64)
    mag = np.abs(p)
    return np.where(mag >= 32768, 0.0, np.sign(p) * 2.0 * mag / 65535)


def _gauss_pcm(seed, n, sigma):
    g = np.random.default_rng(seed).normal(0.0, sigma * 32767, n)
    return np.clip(np.rint(g), -32767, 32767)


def c1_sine(n_frames, freq=1000.0, amp=0.5, fs=48000):
    """C1: mono 1 kHz sine.  -> float64 [(n_frames+1)*HOP] (leading zero hop included)."""
    n = np.arange(n_frames * HOP)
    pcm = np.rint(amp * 32767 * np.sin(2 * np.pi * freq * n / fs))
    return np.concatenate([np.zeros(HOP), pcm_to_float(pcm)])


def c2_noise(n_frames, seed=1234, sigma=0.1):
    """C2: mono Gaussian white noise, sigma = 0.1 full scale."""
    return np.concatenate([np.zeros(HOP), pcm_to_float(_gauss_pcm(seed, n_frames * HOP, sigma))])


def c3_stereo(n_frames, seed_l=1234, seed_r=5678, sigma=0.1):
    """C3: L = g1; R = 0.8 g1 + 0.2 g2 on even hops (M/S wins), 0.1 g2 on odd hops (L/R wins).
    Mixed in the PCM domain and re-quantised to int16 so both channels are genuine 16-bit signals.
    -> float64 [2][(n_frames+1)*HOP]."""
    g1 = _gauss_pcm(seed_l, n_frames * HOP, sigma)
    g2 = _gauss_pcm(seed_r, n_frames * HOP, sigma)
    even = (np.arange(n_frames * HOP) // HOP) % 2 == 0
    r = np.clip(np.rint(np.where(even, 0.8 * g1 + 0.2 * g2, 0.1 * g2)), -32767, 32767)
    z = np.zeros(HOP)
    return np.stack([np.concatenate([z, pcm_to_float(g1)]), np.concatenate([z, pcm_to_float(r)])])


def c4_transients(n_hops, seed=42, floor=0.01, burst=0.5, period=5, short=128):
    """C4: sigma = 0.01 noise floor with a sigma = 0.5 burst of `short` samples at the start of every
    `period`-th hop.  Returns (stream float64 [(n_hops+1)*HOP], list of (offset, a, b) block shapes following the
    forced cycle (1024,1024) -> (1024,128) -> 7x(128,128) -> (128,1024) around each burst hop,
    pacfileThem.py:1192-1210)."""
    rng = np.random.default_rng(seed)
    g = rng.normal(0.0, floor * 32767, n_hops * HOP)
    for h in range(period - 1, n_hops, period):
        g[h * HOP:h * HOP + short] = rng.normal(0.0, burst * 32767, short)
    x = np.concatenate([np.zeros(HOP), pcm_to_float(np.clip(np.rint(g), -32767, 32767))])
    shapes = []
    off, a = 0, HOP
    for h in range(n_hops):
        if h % period == period - 1:
            for _ in range(HOP // short):
                shapes.append((off, a, short)); off += a; a = short
        else:
            shapes.append((off, a, HOP)); off += a; a = HOP
    return x, shapes


def c6_varied(n_frames, seed=6):
    """Mono material with a wide spread of masker levels and slopes inside every frame (what the far-field expansion
    of smr_kernel keys its order on): noise whose level wanders over 60 dB from hop to hop, a few tones of very
    different loudness that come and go, stretches of digital silence and full-scale clipping; 16-bit grid, the
    usual leading hop of zeros.  Not a BASELINE config: a robustness corpus for parity sweeps and tests."""
    rng = np.random.default_rng(seed)
    n = (n_frames + 1) * 1024
    t = np.arange(n)
    level = 10.0 ** (rng.uniform(-4.0, -0.5, n_frames + 1))             # per-hop noise sigma, -80 .. -10 dBFS
    x = rng.normal(0.0, 1.0, n) * np.repeat(level, 1024)
    for _ in range(6):
        f0 = rng.uniform(80.0, 16000.0)
        amp = 10.0 ** rng.uniform(-3.5, -0.3)
        gate = np.repeat(rng.random(n_frames + 1) < 0.6, 1024)
        x += amp * np.sin(2 * np.pi * f0 * t / 48000.0 + rng.uniform(0, 6.28)) * gate
    x[np.repeat(rng.random(n_frames + 1) < 0.05, 1024)] = 0.0            # silence
    loud = np.repeat(rng.random(n_frames + 1) < 0.05, 1024)
    x[loud] *= 30.0                                                      # clips
    x = pcm_to_float(np.clip(np.rint(x * 32767.0), -32767, 32767))
    x[:1024] = 0.0
    return x
