#!/usr/bin/env python3
"""
Host back end throughput (no GPU needed): `.pac` packing of joint stereo long blocks by the C++ packer
(csrc/mrc_pack.cpp), raw / with the Huffman table choice / with the table ids given (as chosen on the device by
huffman_gain_kernel), and the parser.  The blocks are what the encoder produces for the C3 stereo content: a few
hundred frames encoded once by the oracle (CPU), then tiled to the batch size.

    python tools/bench_pack.py [--blocks 16384] [--threads 1,16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd import pacfile as ppac, synth          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16384)
    ap.add_argument("--seed-frames", type=int, default=96)
    ap.add_argument("--threads", default="1,%d" % ppac.get_threads())
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from oracle import fast                                     # only to MAKE realistic encoder outputs on the CPU
    s = synth.c3_stereo(a.seed_frames)
    lvl = 10.0 ** (-1.5 * ((np.arange(s.shape[1]) // 1024) % 5 == 3))      # some quiet hops: Huffman-coded chunks
    s = s * lvl
    bl, br = np.array(fast.blocks_from_stream(s[0], 1024)), np.array(fast.blocks_from_stream(s[1], 1024))
    r = fast.encode_joint_batch(bl, br, 1024, 1024)
    reps = -(-a.blocks // a.seed_frames)
    tile = lambda x: np.ascontiguousarray(np.concatenate([x] * reps)[:a.blocks], dtype=np.int32)   # what the device delivers
    osc, sw, sf, ba, m = (tile(r[k]) for k in ("overall_scale", "ms_switch", "scale_factor", "bit_alloc", "mantissa"))
    m16 = m.astype(np.uint16)                                    # ... or the 16-bit plane of the PCM16 / mantissa16 paths
    cfg = ppac.make_config()
    samples = a.blocks * 1024 * 2
    out = {"blocks": a.blocks, "samples": samples}
    _, _, tables, _ = ppac.pack_joint_blocks(cfg, 1024, 1024, osc, sw, sf, ba, m, use_huffman=True)
    out["huffman_chunks_frac"] = float((tables != 15).mean())

    def timed(fn):
        ts = []
        for _ in range(a.reps):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return float(np.median(ts))

    for nt in [int(v) for v in a.threads.split(",")]:
        ppac.set_threads(nt)
        res = {}
        res["raw"] = timed(lambda: ppac.pack_joint_blocks(cfg, 1024, 1024, osc, sw, sf, ba, m, use_huffman=False))
        res["huffman_priced_on_host"] = timed(lambda: ppac.pack_joint_blocks(cfg, 1024, 1024, osc, sw, sf, ba, m, use_huffman=True))
        res["huffman_tables_given"] = timed(lambda: ppac.pack_joint_blocks(cfg, 1024, 1024, osc, sw, sf, ba, m, huff_table=tables))
        res["huffman_priced_on_host_uint16_plane"] = timed(lambda: ppac.pack_joint_blocks(cfg, 1024, 1024, osc, sw, sf, ba, m16, use_huffman=True))
        data, offs, _, _ = ppac.pack_joint_blocks(cfg, 1024, 1024, osc, sw, sf, ba, m, use_huffman=True)
        buf = ppac.header(cfg, 2, a.blocks * 1024) + data.tobytes()
        cfg2, nch, ns, off = ppac.read_header(buf)
        chunks = ppac.scan_chunks(buf, off)
        res["parse"] = timed(lambda: ppac.unpack_blocks(cfg2, buf, chunks, 2, True))
        out["threads_%d" % nt] = {k: {"s": v, "Msamples_s": samples / v / 1e6} for k, v in res.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
