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


per_shape = {}
for (a, b), o in offs.items():
    enc.h.set_timing(True)
    enc.encode(a, b, x, None, o.numel(), 0, o)
    enc.encode(a, b, x, None, o.numel(), 0, o)
    per_shape["%dx%d" % (a, b)] = {"blocks": int(o.numel()), "stage_ms_mdct_smr_backend": [round(float(v), 4) for v in enc.h.stage_ms()]}
    enc.h.set_timing(False)
dt = timed(run_switched)
print(json.dumps({"workload": "configs[3] block switching (burst every 5th hop)", "hops": hops, "per_shape": per_shape,
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

# ---- decode side: fused dequantise / M-S / IMDCT / window / overlap-add of the joint output above, device resident
Fd = F
o_dev = enc.encode_long(L, R, Fd)
offs_d = (torch.arange(Fd, device=dev, dtype=torch.int64) * 1024).contiguous()
pcm_f = torch.zeros((2, (Fd + 1) * 1024), dtype=torch.float64, device=dev)
pcm_i = torch.empty((2, (Fd + 1) * 1024), dtype=torch.int16, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def run_decode():
    pcm_f.zero_()
    enc.h.dev_decode(1024, 1024, Fd, 2, o_dev["overall_scale"].data_ptr(), o_dev["ms_switch"].data_ptr(),
                     o_dev["scale_factor"].data_ptr(), o_dev["bit_alloc"].data_ptr(), o_dev["mantissa"].data_ptr(),
                     offs_d.data_ptr(), pcm_f[0].data_ptr(), pcm_f[1].data_ptr(), st)
    enc.h.dev_pcm16(pcm_f.numel(), pcm_f.data_ptr(), pcm_i.data_ptr(), st)


dt = timed(run_decode)
err = (pcm_f[0, 2048:Fd * 1024] - L[2048:Fd * 1024])
snr = 10 * torch.log10((L[2048:Fd * 1024] ** 2).sum() / (err ** 2).sum()).item()
print(json.dumps({"workload": "decode: joint stereo long blocks, dequantise..overlap-add + pcm16, device resident",
                  "frames": Fd, "ms_per_step": round(dt * 1e3, 3), "Msamples_per_s": round(2 * Fd * 1024 / dt / 1e6, 1),
                  "algorithmic_GBs": round(Fd * 2 * (4096 + 200 + 8192 * 2 + 2048) / dt / 1e9, 1),
                  "left_channel_snr_db_white_noise": round(snr, 2)}), flush=True)

# ---- host parser: C++ header / chunk parser of a packed stream (single thread)
data, offs_b, _, _ = pacfile.pack_joint_blocks(cfg, 1024, 1024, o["overall_scale"], o["ms_switch"], o["scale_factor"],
                                               o["bit_alloc"], o["mantissa"], True)
blob = pacfile.header(cfg, 2, n * 1024) + data.tobytes()
cfg2, nch, _, off0 = pacfile.read_header(blob)
cfg2.n_short, cfg2.blksw_bits_a, cfg2.blksw_bits_b = 128, 1, 1
t0 = time.perf_counter()
chunks = pacfile.scan_chunks(blob, off0)
parsed = pacfile.unpack_blocks(cfg2, blob, chunks, 2, True)
dt = time.perf_counter() - t0
assert (parsed["mantissa"] == o["mantissa"]).all()
print(json.dumps({"workload": "host unpack_blocks (C++, 1 thread), huffman=True", "frames": n,
                  "Msamples_per_s": round(2 * n * 1024 / dt / 1e6, 1)}), flush=True)

# ---- PCIe-inclusive rate of the per-block host API (mrc_encode_mono: pageable host buffers in, host buffers out)
nb = 16384
xb = noise((nb + 1) * 1024, 99).cpu().numpy()
blocks = np.lib.stride_tricks.sliding_window_view(xb, 2048)[::1024][:nb].copy()
enc.h.encode_mono(blocks[:256], 1024, 1024)
t0 = time.perf_counter()
enc.h.encode_mono(blocks, 1024, 1024)
dt = time.perf_counter() - t0
print(json.dumps({"workload": "host API mrc_encode_mono (explicit blocks from pageable host memory, outputs to host)",
                  "frames": nb, "ms": round(dt * 1e3, 2), "Msamples_per_s": round(nb * 1024 / dt / 1e6, 1),
                  "bytes_over_pcie_per_frame": 16384 + 4096 + 200 + 200 + 8}), flush=True)

# ---- stream mode: many stereo streams advance one block per step, bit reservoirs chained on the device
nS, nT = 2048, 24
gs = torch.Generator(device=dev)
gs.manual_seed(7)
pl = torch.clamp(torch.round(torch.randn((nS, (nT + 1) * 1024), generator=gs, device=dev, dtype=torch.float64) * 3000), -32767, 32767)
sl = (torch.sign(pl) * 2.0 * torch.abs(pl) / 65535).contiguous()
sl[:, :1024] = 0
sr = (0.7 * sl + 0.3 * torch.roll(sl, 17, dims=1)).contiguous()
sr[:, :1024] = 0
shapes_all = [[(i * 1024, 1024, 1024) for i in range(nT)]] * nS
enc.encode_chained(sl, sr, [s[:2] for s in shapes_all])          # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
steps, reservoir = enc.encode_chained(sl, sr, shapes_all)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": "stream mode: %d stereo streams x %d chained joint blocks (Huffman pricing on device)" % (nS, nT),
                  "ms_total": round(dt * 1e3, 2), "ms_per_step": round(dt * 1e3 / nT, 3),
                  "Msamples_per_s": round(2 * nS * nT * 1024 / dt / 1e6, 1),
                  "mean_final_reservoir_bits": round(float(reservoir.double().mean().item()), 1)}), flush=True)
