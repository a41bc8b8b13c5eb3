#!/usr/bin/env python3
"""Experiment: do the HBM-bound kernels of one slice (MDCT, quantiser) overlap the VALU-bound masking kernel of another
when slices of a batch are encoded on two HIP streams (two handles)?  Prints ms per F frames for 1 stream x 1 call and
for 2 streams x unequal slices (so that the streams drift out of phase)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mrcaudiocodec_amd.batch import StreamEncoder

dev = torch.device("cuda", 0)
F = 1 << 17
HOP = 1024
(pcm,) = bench.stream_slice(torch, dev, "c2", 0, F)
encs = [StreamEncoder(device_id=0), StreamEncoder(device_id=0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def one():
    encs[0].encode_long(pcm, None, F, mantissa16=True)

def sliced(sizes_a, sizes_b):
    plan = [(0, s) for s in sizes_a], [(1, s) for s in sizes_b]
    pos = 0
    work = [[], []]
    for k, sizes in enumerate((sizes_a, sizes_b)):
        for s in sizes:
            work[k].append((pos, s)); pos += s
    assert pos == F
    def run():
        for i in range(max(len(work[0]), len(work[1]))):
            for k in (0, 1):
                if i < len(work[k]):
                    f0, n = work[k][i]
                    with torch.cuda.stream(streams[k]):
                        # a distinct output cache key per (stream, slice) would be needed for real use; timing only here
                        encs[k].encode_long(pcm[f0 * HOP:(f0 + n + 1) * HOP], None, n, mantissa16=True)
    return run

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print("one call, one stream: %.3f ms" % timeit(one))
e = F // 8
for name, a, b in (("2 streams, equal halves", [4 * e], [4 * e]),
                   ("2 streams, 1+3+... / 2+2...", [e, 2 * e, e], [2 * e, 2 * e]),
                   ("2 streams, 8 slices staggered", [e // 2, e, e, e, e // 2], [e, e, e, e])):
    print("%-34s %.3f ms" % (name + ":", timeit(sliced(a, b))), flush=True)
