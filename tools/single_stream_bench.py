"""
ONE long stereo stream -> one `.pac` file: the reference's only real use case (pacfileThem.py:1064-1231), measured
both ways on the GPU box:
  before  the block-at-a-time loop (pacfile.encode_stereo_stream_per_block: one mrc_encode_joint + host pack per block,
          the reservoir carried on the host) on the first `--loop-blocks` blocks;
  after   mrc_encode_chained_stream_pcm16_pac on the whole stream (phase A batched, serial scan on the device).
Block shapes come from the transient detector on the content (bursts every `--period` hops).
usage: python tools/single_stream_bench.py [--hops 65536] [--loop-blocks 256] [--period 37]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd import Handle, pacfile, synth, transient      # noqa: E402


def make_stream(hops, period, seed=42):
    """stereo int16 [2][(hops+1)*1024]: noise floor + tone, a burst of 128 samples every `period`-th hop."""
    rng = np.random.default_rng(seed)
    n = hops * 1024
    g1 = rng.normal(0.0, 0.02 * 32767, n)
    g2 = rng.normal(0.0, 0.02 * 32767, n)
    t = np.arange(n)
    tone = 0.2 * 32767 * np.sin(2 * np.pi * 440.0 * t / 48000)
    l = g1 + tone
    r = 0.7 * g1 + 0.3 * g2 + 0.9 * tone
    for h in range(period - 1, hops, period):
        b = rng.normal(0.0, 0.5 * 32767, 128)
        l[h * 1024:h * 1024 + 128] = b
        r[h * 1024:h * 1024 + 128] = 0.8 * b
    pcm = np.zeros((2, (hops + 1) * 1024), np.int16)
    pcm[0, 1024:] = np.clip(np.rint(l), -32767, 32767)
    pcm[1, 1024:] = np.clip(np.rint(r), -32767, 32767)
    return pcm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hops", type=int, default=65536)
    ap.add_argument("--loop-blocks", type=int, default=256)
    ap.add_argument("--period", type=int, default=37)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=0, help="MRC_OPT_CHAIN_THREADS (0: the library's choice)")
    a = ap.parse_args()
    h = Handle(device_id=0)
    h.set_option(4, a.threads)
    pcm = make_stream(a.hops, a.period)
    x = synth.pcm_to_float(pcm)
    t0 = time.perf_counter()
    shapes = transient.block_shapes(h, x)
    t_det = time.perf_counter() - t0
    while shapes and shapes[-1][2] != 1024:
        shapes.pop()
    n_short = sum(1 for (_, aa, bb) in shapes if aa + bb != 2048)
    samples = sum(b for (_, _, b) in shapes) * 2
    # before: the per-block loop on a prefix of the stream that ends with a long block
    k = min(a.loop_blocks, len(shapes))
    while k > 1 and shapes[k - 1][2] != 1024:
        k -= 1
    pre = shapes[:k]
    pacfile.encode_stereo_stream_per_block(h, x, pre[:8] if pre[7][2] == 1024 else pre[:1])      # warm-up
    t0 = time.perf_counter()
    ref = pacfile.encode_stereo_stream_per_block(h, x, pre)
    t_loop = time.perf_counter() - t0
    loop_samples = sum(b for (_, _, b) in pre) * 2
    # after: one call
    got = h.encode_chained_pac(pcm[0][None], pcm[1][None], [pre], num_samples=[sum(b for (_, _, b) in pre)])
    same = got["bytes"].tobytes() == ref
    best = None
    for _ in range(a.reps):
        t0 = time.perf_counter()
        r = h.encode_chained_pac(pcm[0][None], pcm[1][None], [shapes], num_samples=[samples // 2])
        dt = time.perf_counter() - t0
        ms = h.chain_ms()
        if best is None or dt < best[0]:
            best = (dt, ms.tolist(), r["total"])
    dt, ms, total = best
    out = {
        "workload": "one stereo 48 kHz stream, %d hops, burst every %d hops; %d blocks (%d short / transition) from the "
                    "transient detector" % (a.hops, a.period, len(shapes), n_short),
        "before_per_block_loop": {"blocks": len(pre), "ms_per_block": 1e3 * t_loop / len(pre),
                                  "Msamples_s": loop_samples / t_loop / 1e6},
        "after_chained_call": {"blocks": len(shapes), "seconds_host_to_host": dt, "Msamples_s": samples / dt / 1e6,
                               "phase_a_ms": ms[0], "phase_b_ms": ms[1], "pack_ms": ms[2],
                               "phase_b_us_per_block": 1e3 * ms[1] / (len(shapes) + 2), "pac_bytes": total},
        "speedup": (samples / dt) / (loop_samples / t_loop),
        "prefix_bytes_equal_per_block_loop": bool(same),
        "detector_seconds": t_det,
    }
    print(json.dumps(out))
    h.close()


if __name__ == "__main__":
    main()
