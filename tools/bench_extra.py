#!/usr/bin/env python3
"""
Secondary measurements (not the driver's bench line; see bench.py for that): throughput of the other
BASELINE.json configurations on one MI355X, device-resident inputs, plus the host back end.
  configs[2]  stereo 48 kHz through the joint path (M/S decision), long blocks
  configs[3]  block-switching stream: long / start / short / stop shapes mixed (one launch set per shape)
  host        C++ Huffman + bit packer (mrc_pack_*), single host thread, raw and Huffman
Prints one JSON object per measurement.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                                    # noqa: E402
from mrcaudiocodec_amd import pacfile, synth                    # noqa: E402
from mrcaudiocodec_amd.batch import StreamEncoder               # noqa: E402

dev = torch.device("cuda", 0)
enc = StreamEncoder(device_id=0)


def timed(fn, steps=3, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def noise(n, seed, sigma=0.1):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    p = torch.clamp(torch.round(torch.randn((n,), generator=g, device=dev, dtype=torch.float64) * (sigma * 32767)), -32767, 32767)
    return torch.sign(p) * 2.0 * torch.abs(p) / 65535


# ---- configs[2]: stereo, joint path
F = 32768
g1, g2 = noise((F + 1) * 1024, 1234), noise((F + 1) * 1024, 5678)
even = ((torch.arange((F + 1) * 1024, device=dev) // 1024) % 2 == 0)
L = g1.contiguous()
R = torch.where(even, 0.8 * g1 + 0.2 * g2, 0.1 * g2).contiguous()
dt = timed(lambda: enc.encode_long(L, R, F))
out = enc.encode_long(L, R, F)
print(json.dumps({"workload": "configs[2] stereo joint M/S, long blocks", "frames": F, "ms_per_step": round(dt * 1e3, 3),
                  "Msamples_per_s": round(2 * F * 1024 / dt / 1e6, 1),
                  "ms_switch_on_fraction": round(float(out["ms_switch"].double().mean().item()), 3)}), flush=True)

# ---- configs[3]: block switching, mono stream, one burst every 5th hop
hops = 20000
x_np, shapes = synth.c4_transients(hops)
x = torch.from_numpy(x_np).to(dev)
by_shape = {}
for (o, a, b) in shapes:
    by_shape.setdefault((a, b), []).append(o)
offs = {k: torch.tensor(v, dtype=torch.int64, device=dev) for k, v in by_shape.items()}


def run_switched():
    for (a, b), o in offs.items():
        if o.numel() % 2 == 0 or True:
            enc.encode(a, b, x, None, o.numel(), 0, o)


dt = timed(run_switched)
print(json.dumps({"workload": "configs[3] block switching (burst every 5th hop)", "hops": hops,
                  "blocks": {"%dx%d" % k: int(v.numel()) for k, v in offs.items()},
                  "ms_per_step": round(dt * 1e3, 3), "Msamples_per_s": round(hops * 1024 / dt / 1e6, 1)}), flush=True)

# ---- host back end: C++ Huffman + bit packing of the joint output (single thread)
n = 4096
o = {k: v[:n].cpu().numpy() for k, v in out.items()}
cfg = pacfile.make_config()
for huff in (False, True):
    t0 = time.perf_counter()
    data, offs_b, table, saved = pacfile.pack_joint_blocks(cfg, 1024, 1024, o["overall_scale"], o["ms_switch"],
                                                            o["scale_factor"], o["bit_alloc"], o["mantissa"], huff)
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "host pack_joint_blocks (C++, 1 thread), huffman=%s" % huff, "frames": n,
                      "Msamples_per_s": round(2 * n * 1024 / dt / 1e6, 1), "bytes_per_frame": round(len(data) / n, 1),
                      "kbit_per_s_at_48k": round(len(data) * 8 / (n * 1024 / 48000) / 1e3, 1)}), flush=True)
