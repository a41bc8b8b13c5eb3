#!/usr/bin/env python3
"""
Large parity sweep on the GPU box: the HIP path (through the C ABI) against the batched oracle on many
seeded frames, counting every integer output that differs.  The bar is zero; if an entry ever differs the
sweep prints where and how close the deciding float64 values were (see DESIGN.md "Parity policy").

    python tools/parity_sweep.py [--mono 16384] [--joint 4096] [--chunk 1024] > profiles/<tag>_parity_sweep.txt
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrcaudiocodec_amd import Handle, synth       # noqa: E402
from oracle import fast                            # noqa: E402  (the checker)

INT_KEYS = ("overall_scale", "bit_alloc", "scale_factor", "mantissa", "reservoir_out")


def compare(got, ref, keys, base):
    bad_frames = set()
    bad_entries = 0
    for k in keys:
        d = np.asarray(got[k]) != np.asarray(ref[k])
        if d.any():
            idx = np.argwhere(d)
            bad_entries += len(idx)
            bad_frames.update(int(i[0]) + base for i in idx)
            print("  MISMATCH %s: %d entries, first frame %d, got %s want %s" %
                  (k, len(idx), int(idx[0][0]) + base, np.asarray(got[k])[tuple(idx[0])], np.asarray(ref[k])[tuple(idx[0])]))
    return bad_frames, bad_entries


def varied_stream(n_frames, seed):
    return synth.c6_varied(n_frames, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mono", type=int, default=16384)
    ap.add_argument("--joint", type=int, default=4096)
    ap.add_argument("--varied", type=int, default=0, help="frames of the varied-level corpus (mono and as L/R pair)")
    ap.add_argument("--shapes", type=int, default=0, help="blocks per short / transition shape (mono, and a quarter joint)")
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--seed-offset", type=int, default=0, help="added to every generator seed: a sweep over OTHER frames than the default's")
    ap.add_argument("--exact-spread", action="store_true",
                    help="MRC_OPT_EXACT_SPREAD: the spreading function operation by operation in masker order (psychoac.py:68-78)")
    args = ap.parse_args()
    h = Handle()
    if args.exact_spread:
        h.set_option(1, 1)
        print("# MRC_OPT_EXACT_SPREAD = %d" % h.get_option(1), flush=True)
    t0 = time.time()
    for (a, b) in ((128, 128), (1024, 128), (128, 1024)) if args.shapes else ():
        for joint, n in ((False, args.shapes), (True, args.shapes // 4)):
            frames_bad, entries_bad, mdct_err = set(), 0, 0.0
            rng = np.random.default_rng(a * 31 + b + int(joint) + args.seed_offset)
            for base in range(0, n, args.chunk * 4):
                c = min(args.chunk * 4, n - base)
                sig = 10.0 ** rng.uniform(-3.0, -0.3, (c, 1))          # per-block level, -60 .. -6 dBFS
                mk = lambda: synth.pcm_to_float(np.clip(np.rint(rng.normal(0, 1, (c, a + b)) * sig * 32767), -32767, 32767))
                bl = mk()
                res_in = rng.integers(-100, 200, c)
                if joint:
                    br = np.where((np.arange(c) % 2 == 0)[:, None], 0.9 * bl + 0.1 * mk(), mk())
                    got = h.encode_joint(bl, br, a, b, res_in, want_mdct=True)
                    ref = fast.encode_joint_batch(bl, br, a, b, res_in)
                    keys = INT_KEYS + ("ms_switch",)
                else:
                    got = h.encode_mono(bl, a, b, res_in, want_mdct=True)
                    ref = fast.encode_mono_batch(bl, a, b, res_in)
                    keys = INT_KEYS
                fb, eb = compare(got, ref, keys, base)
                frames_bad |= fb
                entries_bad += eb
                mdct_err = max(mdct_err, float(np.abs(got["mdct"] - ref["mdct"]).max() / max(np.abs(ref["mdct"]).max(), 1e-300)))
                print("shape %dx%d %s: %d/%d blocks done, %d mismatching so far (%.0f s)" %
                      (a, b, "joint" if joint else "mono", base + c, n, len(frames_bad), time.time() - t0), flush=True)
            print("RESULT shape %dx%d %s noise of varying level: blocks=%d mismatching_blocks=%d mismatching_entries=%d max_rel_mdct_err=%.3g" %
                  (a, b, "joint" if joint else "mono", n, len(frames_bad), entries_bad, mdct_err), flush=True)
    for name, n, joint in (("mono varied levels/tones/silence/clipping", args.varied, False),
                           ("joint varied levels/tones/silence/clipping", args.varied // 4, True)):
        frames_bad, entries_bad, mdct_err = set(), 0, 0.0
        rng = np.random.default_rng(7)
        for base in range(0, n, args.chunk):
            c = min(args.chunk, n - base)
            res_in = rng.integers(-300, 800, c)
            xl = varied_stream(c, 500000 + args.seed_offset + base)
            bl = np.array(fast.blocks_from_stream(xl, 1024))
            if joint:
                xr = 0.6 * xl + 0.4 * varied_stream(c, 900000 + args.seed_offset + base)
                br = np.array(fast.blocks_from_stream(xr, 1024))
                got = h.encode_joint(bl, br, 1024, 1024, res_in, want_mdct=True)
                ref = fast.encode_joint_batch(bl, br, 1024, 1024, res_in)
                keys = INT_KEYS + ("ms_switch",)
            else:
                got = h.encode_mono(bl, 1024, 1024, res_in, want_mdct=True)
                ref = fast.encode_mono_batch(bl, 1024, 1024, res_in)
                keys = INT_KEYS
            fb, eb = compare(got, ref, keys, base)
            frames_bad |= fb
            entries_bad += eb
            mdct_err = max(mdct_err, float(np.abs(got["mdct"] - ref["mdct"]).max() / max(np.abs(ref["mdct"]).max(), 1e-300)))
            print("%s: %d/%d frames done, %d mismatching frames so far (%.0f s)" %
                  (name, base + c, n, len(frames_bad), time.time() - t0), flush=True)
        if n:
            print("RESULT %s: frames=%d mismatching_frames=%d mismatching_entries=%d max_rel_mdct_err=%.3g" %
                  (name, n, len(frames_bad), entries_bad, mdct_err), flush=True)
    for name, n, joint in (("mono C2 noise", args.mono, False), ("joint C3 stereo", args.joint, True)):
        frames_bad, entries_bad, mdct_err = set(), 0, 0.0
        rng = np.random.default_rng(2026)
        for base in range(0, n, args.chunk):
            c = min(args.chunk, n - base)
            seed = 100000 + args.seed_offset + base
            res_in = rng.integers(-300, 800, c)
            if joint:
                s = synth.c3_stereo(c, seed_l=seed, seed_r=seed + 1)
                bl, br = np.array(fast.blocks_from_stream(s[0], 1024)), np.array(fast.blocks_from_stream(s[1], 1024))
                got = h.encode_joint(bl, br, 1024, 1024, res_in, want_mdct=True)
                ref = fast.encode_joint_batch(bl, br, 1024, 1024, res_in)
                keys = INT_KEYS + ("ms_switch",)
            else:
                blocks = np.array(fast.blocks_from_stream(synth.c2_noise(c, seed=seed), 1024))
                got = h.encode_mono(blocks, 1024, 1024, res_in, want_mdct=True)
                ref = fast.encode_mono_batch(blocks, 1024, 1024, res_in)
                keys = INT_KEYS
            fb, eb = compare(got, ref, keys, base)
            frames_bad |= fb
            entries_bad += eb
            mdct_err = max(mdct_err, float(np.abs(got["mdct"] - ref["mdct"]).max() / np.abs(ref["mdct"]).max()))
            print("%s: %d/%d frames done, %d mismatching frames so far (%.0f s)" %
                  (name, base + c, n, len(frames_bad), time.time() - t0), flush=True)
        print("RESULT %s: frames=%d mismatching_frames=%d mismatching_entries=%d max_rel_mdct_err=%.3g" %
              (name, n, len(frames_bad), entries_bad, mdct_err), flush=True)
    h.close()


if __name__ == "__main__":
    main()
