"""SHA-256 of the MDCT lines and of every integer output for a fixed set of long-block layouts (hop-overlapped streams,
explicit offsets: a hop apart / in runs / at odd sample offsets, blocks at stride 2048; mono and joint; int16 and float64
samples; ragged counts).  Run it with two builds of the library (MRC_HIP_LIBRARY=...) and diff the outputs: a rework of
mdct_long_kernel that keeps every floating-point operation and its order must give the same digests.
usage: python tools/mdct_bitwise.py > out.json"""
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


def case(name, left, right, n, stride, offsets):
    nsig = 4 if right is not None else 1
    lines = torch.full((n * nsig * 1024,), float("nan"), dtype=torch.float64, device=dev)
    out = enc.encode(1024, 1024, left, right, n, stride, offsets, lines_out=lines, fresh=True, offsets_checked=offsets is not None)
    torch.cuda.synchronize()
    d = {"case": name, "lines": digest(lines)}
    for k in sorted(out):
        d[k] = digest(out[k])
    print(json.dumps(d), flush=True)


ar = torch.arange(4099, device=dev, dtype=torch.int64)
offs_hop = (ar * 1024).contiguous()
offs_runs = ((ar // 4) * 5 + (ar % 4)) * 1024            # runs of four, then a hop skipped
offs_runs = offs_runs[offs_runs < (HOPS - 2) * 1024].contiguous()
offs_odd = (offs_runs + 333).contiguous()                # odd sample offsets
perm = torch.randperm(2051, generator=torch.Generator().manual_seed(3)).to(dev)
offs_rand = (perm * 2048 + (perm % 7)).contiguous()      # unordered, mixed parity, never a hop apart
for (tag, L, R) in (("i16", pl, pr), ("f64", fl, fr)):
    case(tag + " mono stream 4099", L, None, 4099, 1024, None)
    case(tag + " joint stream 2051", L, R, 2051, 1024, None)
    case(tag + " mono offsets a hop apart", L, None, offs_hop.numel(), 0, offs_hop)
    case(tag + " joint offsets a hop apart", L, R, 1027, 0, offs_hop[:1027].contiguous())
    case(tag + " mono offsets runs of 4", L, None, offs_runs.numel(), 0, offs_runs)
    case(tag + " joint offsets runs of 4", L, R, 1001, 0, offs_runs[:1001].contiguous())
    case(tag + " mono odd offsets", L, None, offs_odd.numel(), 0, offs_odd)
    case(tag + " joint odd offsets", L, R, 999, 0, offs_odd[:999].contiguous())
    case(tag + " mono unordered offsets", L, None, 2051, 0, offs_rand)
    case(tag + " joint unordered offsets", L, R, 777, 0, offs_rand[:777].contiguous())
    case(tag + " mono blocks at stride 2048", L, None, 2400, 2048, None)
    case(tag + " joint blocks at stride 2048", L, R, 1203, 2048, None)
    case(tag + " mono base on an odd sample", L[1:], None, 1500, 1024, None)
    case(tag + " joint bases on odd samples", L[1:], R[3:], 1500, 1024, None)
