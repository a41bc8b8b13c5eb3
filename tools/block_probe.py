#!/usr/bin/env python3
"""Where the time of ONE block through the drop-in seam goes (mrc_encode_joint at n = 1 + the host packer), on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrcaudiocodec_amd import Handle, synth, pacfile as ppac       # noqa: E402

h = Handle()
c = h.cfg
cfg = ppac.make_config(c.sample_rate, c.n_mdct_lines, c.n_short, c.n_scale_bits, c.n_mant_size_bits,
                       c.target_bits_per_sample, c.blksw_bits_a, c.blksw_bits_b)
xs = synth.c3_stereo(64)
N = 200
for (a, b) in ((1024, 1024), (128, 128), (1024, 128)):
    bl = [xs[0, i * 1024:i * 1024 + a + b][None, :].copy() for i in range(8)]
    br = [xs[1, i * 1024:i * 1024 + a + b][None, :].copy() for i in range(8)]
    r = h.encode_joint(bl[0], br[0], a, b, [0])
    t0 = time.perf_counter()
    for i in range(N):
        r = h.encode_joint(bl[i % 8], br[i % 8], a, b, [0])
    t_enc = (time.perf_counter() - t0) / N
    h.set_timing(True)
    km = np.zeros(5)
    for i in range(20):
        h.encode_joint(bl[i % 8], br[i % 8], a, b, [0])
        km += h.kernel_ms()
    h.set_timing(False)
    t0 = time.perf_counter()
    for i in range(N):
        ppac.pack_joint_blocks(cfg, a, b, r["overall_scale"], r["ms_switch"], r["scale_factor"], r["bit_alloc"], r["mantissa"], True)
    t_pack = (time.perf_counter() - t0) / N
    print("shape %4d+%4d: encode_joint %.1f us per call (kernels on the device: %s = %.1f us), host pack %.1f us"
          % (a, b, t_enc * 1e6, np.round(km / 20 * 1e3, 1).tolist(), km.sum() / 20 * 1e3, t_pack * 1e6), flush=True)
