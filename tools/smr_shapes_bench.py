"""Device time of smr_kernel (and the MDCT in front of it) per block shape, mono int16 PCM, at the unit counts a block-switched
stream of 131 072 hops with a transient every ~9th hop produces (hipEvents of the library, mrc_set_timing).
usage: python tools/smr_shapes_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd.batch import StreamEncoder      # noqa: E402

F = 131072
enc = StreamEncoder(device_id=0)
dev = enc.device
g = torch.Generator(device=dev)
g.manual_seed(1)
pl = torch.clamp(torch.round(torch.randn(((F + 1) * 1024,), generator=g, device=dev, dtype=torch.float64) * 3000), -32767, 32767).to(torch.int16)


def stage_ms(fn, reps=5):
    enc.h.set_timing(True)
    fn()
    acc = [0.0, 0.0]
    for _ in range(reps):
        fn()
        k = enc.h.kernel_ms()
        acc[0] += k[0]
        acc[1] += k[1]
    enc.h.set_timing(False)
    return acc[0] / reps, acc[1] / reps


for (a, b, n, step) in ((128, 128, 114688, 128), (1024, 128, 14336, 1024), (128, 1024, 14336, 1024), (1024, 1024, 131072, 1024)):
    n = min(n, (pl.numel() - a - b) // step)
    offs = (torch.arange(n, device=dev, dtype=torch.int64) * step).contiguous()
    m, s = stage_ms(lambda: enc.encode(a, b, pl, None, n, 0, offs, mantissa16=True, offsets_checked=True))
    print(json.dumps({"shape": [a, b], "units": n, "mdct_ms": round(m, 4), "smr_ms": round(s, 4)}))
