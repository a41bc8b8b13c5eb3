#!/usr/bin/env python3
"""
Which evaluation of the upper-side spreading sum smr_kernel's long-block units / chunks take, per corpus (needs the
-DMRC_NODE_STATS build: make OUT=.../libmrc_hip_nodestats.so BUILD=build_nodestats EXTRA=-DMRC_NODE_STATS).
    MRC_HIP_LIBRARY=mrcaudiocodec_amd/libmrc_hip_nodestats.so python tools/node_stats.py [frames]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrcaudiocodec_amd import Handle, synth, _lib        # noqa: E402
from oracle.fast import blocks_from_stream               # noqa: E402  (stream -> overlapped blocks only)


def stats(reset=True):
    out = (ctypes.c_ulonglong * 4)()
    assert _lib.lib.mrc_debug_node_stats(out, 1 if reset else 0) == 0
    return list(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    h = Handle()
    stats()
    rng = np.random.default_rng(5)

    def cliff(cut, amp, floor=True):
        g = rng.normal(0, 1, (n + 1) * 1024)
        G = np.fft.rfft(g); G[int(len(G) * cut / 24000):] = 0; g = np.fft.irfft(G, len(g))
        g = g / g.std() * amp
        x = synth.pcm_to_float(np.clip(np.rint(g * 32767), -32767, 32767)) if floor else g
        x[:1024] = 0
        return x
    corpora = [("noise sigma 0.1", synth.c2_noise(n)), ("noise sigma 0.001", synth.c2_noise(n, sigma=0.001)),
               ("noise sigma 0.3", synth.c2_noise(n, sigma=0.3)), ("varied", synth.c6_varied(n)),
               ("sine", synth.c1_sine(n)), ("lowpass 9 kHz", cliff(9000, 0.05)), ("lowpass 4 kHz", cliff(4000, 0.25)),
               ("lowpass 4 kHz, float", cliff(4000, 0.05, False)), ("transients (long blocks)", synth.c4_transients(n)[0])]
    for name, x in corpora:
        h.encode_mono(blocks_from_stream(x, 1024, n), 1024, 1024)
        s = stats()
        print("%-26s units: nodes %6d  sorted sweep %6d | chunks of node units: by nodes %7d  sent back %6d (%.3f %%)"
              % (name, s[0] // 4, s[1] // 4, s[2], s[3], 100.0 * s[3] / max(s[2] + s[3], 1)))
    xs = synth.c3_stereo(n)
    h.encode_joint(blocks_from_stream(xs[0], 1024, n), blocks_from_stream(xs[1], 1024, n), 1024, 1024)
    s = stats()
    print("%-26s units: nodes %6d  sorted sweep %6d | chunks of node units: by nodes %7d  sent back %6d (%.3f %%)"
          % ("C3 stereo (joint)", s[0] // 4, s[1] // 4, s[2], s[3], 100.0 * s[3] / max(s[2] + s[3], 1)))


if __name__ == "__main__":
    main()
