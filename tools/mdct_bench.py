"""Device time of the MDCT kernel alone (hipEvents of the library, mrc_set_timing) for the layouts the encoder feeds it:
hop-overlapped streams (mono / joint), explicit offsets a hop apart (what a block-switched stream and the chained encode
hand over for their runs of long blocks), and the short / transition shapes.  int16 PCM in.
usage: python tools/mdct_bench.py [frames]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd.batch import StreamEncoder      # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
enc = StreamEncoder(device_id=0)
dev = enc.device
g = torch.Generator(device=dev)
g.manual_seed(1)
pl = torch.clamp(torch.round(torch.randn(((F + 1) * 1024,), generator=g, device=dev, dtype=torch.float64) * 3000), -32767, 32767).to(torch.int16)
pr = torch.roll(pl, 333)
HBM = 8000.0


def mdct_ms(fn, reps=5):
    enc.h.set_timing(True)
    fn()
    acc = 0.0
    for _ in range(reps):
        fn()
        acc += enc.h.kernel_ms()[0]
    enc.h.set_timing(False)
    return acc / reps


rows = []


def row(name, ms, units, bytes_per_unit):
    gbs = units * bytes_per_unit / (ms * 1e-3) / 1e9
    rows.append({"case": name, "ms": round(ms, 4), "units": units, "GBs": round(gbs, 1), "frac_hbm": round(gbs / HBM, 4)})


offs = torch.arange(F, device=dev, dtype=torch.int64) * 1024
Fj = F // 2
offs_j = offs[:Fj].contiguous()
row("mono stream (stride 1024)", mdct_ms(lambda: enc.encode_long(pl, None, F, mantissa16=True)), F, 2048 + 8192)
row("mono offsets a hop apart", mdct_ms(lambda: enc.encode(1024, 1024, pl, None, F, 0, offs, mantissa16=True, offsets_checked=True)), F, 2048 + 8192)
# a block-switched stream: runs of 4 long blocks, then a gap of one hop (where the short blocks would sit)
run = (torch.arange(F, device=dev, dtype=torch.int64) // 4) * 5 + (torch.arange(F, device=dev, dtype=torch.int64) % 4)
run = (run[run < F - 2] * 1024).contiguous()
row("mono offsets, runs of 4", mdct_ms(lambda: enc.encode(1024, 1024, pl, None, run.numel(), 0, run, mantissa16=True, offsets_checked=True)), run.numel(), 2048 + 8192)
row("joint stream (stride 1024)", mdct_ms(lambda: enc.encode_long(pl, pr, Fj, mantissa16=True)), Fj, 2 * 2048 + 4 * 8192)
row("joint offsets a hop apart", mdct_ms(lambda: enc.encode(1024, 1024, pl, pr, Fj, 0, offs_j, mantissa16=True, offsets_checked=True)), Fj, 2 * 2048 + 4 * 8192)
# short and transition shapes, mono and joint, at the offsets a transient every 5th hop produces
n_s = F // 8
o_short = (torch.arange(n_s, device=dev, dtype=torch.int64) * 128).contiguous()
row("mono short (128,128)", mdct_ms(lambda: enc.encode(128, 128, pl, None, n_s, 0, o_short, mantissa16=True, offsets_checked=True)), n_s, 256 + 1024)
row("joint short (128,128)", mdct_ms(lambda: enc.encode(128, 128, pl, pr, n_s, 0, o_short, mantissa16=True, offsets_checked=True)), n_s, 2 * 256 + 4 * 1024)
n_t = F // 16
o_tr = (torch.arange(n_t, device=dev, dtype=torch.int64) * 5120).contiguous()
row("mono start (1024,128)", mdct_ms(lambda: enc.encode(1024, 128, pl, None, n_t, 0, o_tr, mantissa16=True, offsets_checked=True)), n_t, 256 + 4608)
row("mono stop (128,1024)", mdct_ms(lambda: enc.encode(128, 1024, pl, None, n_t, 0, o_tr, mantissa16=True, offsets_checked=True)), n_t, 2048 + 4608)
row("joint start (1024,128)", mdct_ms(lambda: enc.encode(1024, 128, pl, pr, n_t, 0, o_tr, mantissa16=True, offsets_checked=True)), n_t, 2 * 256 + 4 * 4608)
for r in rows:
    print(json.dumps(r))
