#!/usr/bin/env python3
"""
End-to-end byte parity on many streams (GPU box): for S synthetic stereo streams with transients, tones and noise of
varying level -- transient detector -> block shapes -> chained joint encode with Huffman -> `.pac` bytes -- the
product path (GPU kernels + C++ packer, all streams advanced together with the reservoirs chained on the device)
against the oracle (one stream at a time on the CPU), then decode on the GPU against the oracle's decoder.

    python tools/pac_sweep.py [--streams 16] [--hops 48] > profiles/<tag>_pac_sweep.txt
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrcaudiocodec_amd import Handle, pacfile, synth, transient          # noqa: E402
from oracle import codec as ocodec, decode as odec, pacfile as opac, transient as otrans   # noqa: E402  (the checker)


def make_stream(hops, seed):
    rng = np.random.default_rng(seed)
    base = synth.c6_varied(hops, seed=seed)
    n = len(base)
    burst = np.zeros(n)
    for h in rng.choice(np.arange(2, hops - 4), size=max(1, hops // 6), replace=False):      # clicks -> block switching (not at the very end: Close() needs a long last block)
        pos = h * 1024 + int(rng.integers(0, 900))
        burst[pos:pos + 96] += rng.normal(0, 0.4, 96)
    left = base + burst
    right = 0.6 * left + 0.4 * synth.c6_varied(hops, seed=seed + 10000)
    s = np.stack([left, right])
    # a steady ending (tone + low noise over the last five hops): the reference's Close() needs a long last block
    tail = slice((hops - 4) * 1024, n)
    t = np.arange(n)[tail]
    s[:, tail] = 0.2 * np.sin(2 * np.pi * 440.0 * t / 48000.0) + rng.normal(0, 1e-3, (2, t.size))
    s = synth.pcm_to_float(np.clip(np.rint(s * 32767), -32767, 32767))
    s[:, :1024] = 0.0
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=16)
    ap.add_argument("--hops", type=int, default=48)
    ap.add_argument("--seed", type=int, default=1000, help="seed of the first stream (stream s takes seed + s)")
    a = ap.parse_args()
    h = Handle()
    t0 = time.time()
    streams = np.stack([make_stream(a.hops, a.seed + s) for s in range(a.streams)])
    shapes = [transient.block_shapes(h, streams[s]) for s in range(a.streams)]
    got = pacfile.encode_stereo_streams(h, streams, shapes, use_huffman=True)
    bad_shapes = bad_bytes = bad_dec = 0
    n_blocks = n_short = 0
    worst = 0.0
    for s in range(a.streams):
        cp = ocodec.default_params(nChannels=2)
        want_shapes = otrans.block_shapes(streams[s], cp)
        if list(map(tuple, want_shapes)) != list(map(tuple, shapes[s])):
            bad_shapes += 1
        want = opac.encode_stereo_stream(streams[s], want_shapes, huffman=True)
        if want != got[s]:
            bad_bytes += 1
        n_blocks += len(want_shapes)
        n_short += sum(1 for (_, x, y) in want_shapes if (x, y) != (1024, 1024))
        _, ref = odec.decode_pac(want)
        _, dec = pacfile.decode_pac(h, got[s])
        err = float(np.abs(dec.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-300))
        worst = max(worst, err)
        if err > 1e-12 or not np.array_equal(pacfile.decode_pac_pcm16(h, got[s]), odec.pcm16(ref[:, 1024:])):
            bad_dec += 1
        print("stream %d/%d: %d blocks (%d not long), %d bytes, %.0f s" %
              (s + 1, a.streams, len(want_shapes), sum(1 for (_, x, y) in want_shapes if (x, y) != (1024, 1024)),
               len(want), time.time() - t0), flush=True)
    print("RESULT pac sweep: streams=%d hops=%d seed=%d blocks=%d (not long: %d) shape_mismatches=%d byte_mismatches=%d "
          "decode_mismatches=%d max_rel_decode_err=%.3g" % (a.streams, a.hops, a.seed, n_blocks, n_short, bad_shapes, bad_bytes,
                                                             bad_dec, worst),
          flush=True)
    h.close()


if __name__ == "__main__":
    main()
