"""SHA-256 of the MDCT lines and of every integer output for the SHORT and TRANSITION block shapes (mdct_wave_kernel):
(128,128), (1024,128), (128,1024); mono and joint; int16 and float64 samples; explicit offsets (ordered, odd, unordered)
and strides; counts from one block to several rounds of workgroups, ragged.  Run it with two builds of the library
(MRC_HIP_LIBRARY=...) and diff the outputs: a rework of the kernel that keeps every floating-point operation and its order
must give the same digests (the long block's twin is tools/mdct_bitwise.py).
usage: python tools/mdct_bitwise_shapes.py > out.json"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd.batch import StreamEncoder      # noqa: E402

enc = StreamEncoder(device_id=0)
dev = enc.device
g = torch.Generator(device=dev)
g.manual_seed(20260405)
HOPS = 5000
pl = torch.clamp(torch.round(torch.randn(((HOPS + 2) * 1024,), generator=g, device=dev, dtype=torch.float64) * 5000), -32768, 32767).to(torch.int16)
pr = torch.clamp(torch.round(pl.to(torch.float64) * 0.7 + torch.randn(pl.shape, generator=g, device=dev, dtype=torch.float64) * 900), -32768, 32767).to(torch.int16)
pl[1024:1040] = -32768                                   # the code without a positive twin
fl = (pl.to(torch.float64) / 32767.0).contiguous()
fr = (pr.to(torch.float64) / 32767.0).contiguous()


def digest(t):
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:24]


def case(name, a, b, left, right, n, stride, offsets):
    nsig = 4 if right is not None else 1
    lines = torch.full((n * nsig * ((a + b) // 2),), float("nan"), dtype=torch.float64, device=dev)
    out = enc.encode(a, b, left, right, n, stride, offsets, lines_out=lines, fresh=True, offsets_checked=offsets is not None)
    torch.cuda.synchronize()
    d = {"case": name, "lines": digest(lines)}
    for k in sorted(out):
        d[k] = digest(out[k])
    print(json.dumps(d), flush=True)


gen = torch.Generator().manual_seed(11)
for (a, b) in ((128, 128), (1024, 128), (128, 1024)):
    n_win = a + b
    span = (HOPS * 1024 - n_win - 8)
    for (tag, L, R) in (("i16", pl, pr), ("f64", fl, fr)):
        for n in (1, 3, 5, 63, 64, 65, 1000, 4097, 26214, 30001):
            step = max(1, min(n_win // 2, span // n))
            offs = (torch.arange(n, dtype=torch.int64) * step).to(dev).contiguous()
            case("%s %dx%d mono ordered offsets n=%d" % (tag, a, b, n), a, b, L, None, n, 0, offs)
            if n <= 4097:
                case("%s %dx%d joint ordered offsets n=%d" % (tag, a, b, n), a, b, L, R, n, 0, offs)
        offs_odd = (torch.arange(2500, dtype=torch.int64) * 1333 + 7).to(dev).contiguous()
        case("%s %dx%d mono odd offsets" % (tag, a, b), a, b, L, None, 2500, 0, offs_odd)
        case("%s %dx%d joint odd offsets" % (tag, a, b), a, b, L, R, 2500, 0, offs_odd)
        perm = torch.randperm(3001, generator=gen)
        offs_rand = (perm * 1100 + (perm % 5)).to(dev).contiguous()
        case("%s %dx%d mono unordered offsets" % (tag, a, b), a, b, L, None, 3001, 0, offs_rand)
        case("%s %dx%d joint unordered offsets" % (tag, a, b), a, b, L, R, 777, 0, offs_rand[:777].contiguous())
        case("%s %dx%d mono blocks at stride %d" % (tag, a, b, n_win // 2), a, b, L, None, 2400, n_win // 2, None)
        case("%s %dx%d joint blocks at stride %d" % (tag, a, b, n_win), a, b, L, R, 1203, n_win, None)
        case("%s %dx%d joint bases on odd samples" % (tag, a, b), a, b, L[1:], R[3:], 1500, 640, None)
