#!/usr/bin/env python3
"""Host-to-host rate of mrc_encode_stream_pcm16 (page-locked int16 PCM -> codes in page-locked memory) against the
chunk size and the length of the stream.  Usage: python tools/h2h_sweep.py [frames ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from mrcaudiocodec_amd import PinnedArray
from mrcaudiocodec_amd.batch import StreamEncoder

HOP, NB = 1024, 25
dev = torch.device("cuda", 0)
enc = StreamEncoder(device_id=0)
for F in [int(a) for a in sys.argv[1:]] or [131072, 524288]:
    keep = []
    def pin(shape, dt):
        p = PinnedArray(shape, dt); keep.append(p); return p.array
    host_pcm = pin(((F + 1) * HOP,), np.int16)
    step = 131072
    for f0 in range(0, F + 1, step):
        n = min(step, F + 1 - f0)
        (pcm,) = bench.stream_slice(torch, dev, "c2", f0, n if f0 + n < F + 1 else n - 1)
        host_pcm[f0 * HOP:f0 * HOP + pcm.numel()] = pcm.cpu().numpy()
    outs = dict(overall_scale=pin((F, 1), np.int32), scale_factor=pin((F, 1, NB), np.int32),
                bit_alloc=pin((F, 1, NB), np.int32), mantissa=pin((F, 1, HOP), np.uint16), reservoir_out=pin((F,), np.int32))
    for chunk in (8192, 16384, 32768, 65536):
        enc.h.encode_stream_pcm16(host_pcm, None, None, chunk, outs)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            enc.h.encode_stream_pcm16(host_pcm, None, None, chunk, outs)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        print("frames %8d chunk %6d: %8.1f Msamples/s (%.1f GB/s each way)" % (F, chunk, F * HOP / t / 1e6, F * 2 * HOP / t / 1e9), flush=True)
    for p in keep:
        p.free()
