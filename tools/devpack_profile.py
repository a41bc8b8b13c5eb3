#!/usr/bin/env python3
"""Encode a batch of stereo frames once, then pack it on the device a few times (for rocprofv3 --kernel-trace --stats:
the pack_* kernels' durations).  Usage: python3 tools/devpack_profile.py [stereo_frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mrcaudiocodec_amd.batch import StreamEncoder

Fs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
enc = StreamEncoder(device_id=0)
sl, sr = bench.stream_slice(torch, dev, "c3", 0, Fs)
out = enc.encode_long(sl, sr, Fs, mantissa16=True)
for use in (True, False, True, False, True, False):
    p = enc.pack(1024, 1024, out, use_huffman=use)
torch.cuda.synchronize()
print("stereo frames", Fs, "bytes per frame", p["bytes"].numel() / Fs)
