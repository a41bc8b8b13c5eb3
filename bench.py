#!/usr/bin/env python3
"""
bench.py -- encode throughput of the MI355X hot path (BASELINE.json metric: encode Msamples/s,
48 kHz, 2048-point MDCT blocks).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (N > 1, one rank per GPU)

A step = one pass of the whole hot path (window+MDCT -> masked threshold/SMR -> bit allocation ->
scale factors + mantissas) over one batch of synthetic PCM frames that is already resident in HBM.
Workload at every N: BASELINE.json configs[1] -- mono 48 kHz Gaussian white noise (sigma 0.1 FS,
16-bit PCM grid), all-long blocks (a=b=1024), independent-frames mode (reservoir_in = 0), F frames per
GPU per step in the hop-overlapped stream layout.  Frames shard across ranks as disjoint streams with
no collective ("weak" scaling: per-GPU work fixed).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant kernel (largest share of device time): algorithmic bytes per launch / its
                average duration measured with hipEvents on the launch stream, against 8 TB/s HBM.
  kernels       the same for every kernel of the path (the MDCT kernel is the one north_star prices
                against the HBM roofline; the SMR kernel is fp64-VALU bound).
  cpu_baseline  the oracle's faithful NumPy port of the reference path, 1 core, bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOP = 1024
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
# algorithmic bytes per long (frame, channel) -- DESIGN.md "Algorithmic bytes"
BYTES_MDCT = 8192 + 8192                          # one new hop in (f64) + 1024 lines out (f64)
BYTES_SMR = 8192 + 8192 + 4 + 25 * 8 + 25 * 8     # hop in + lines in + overall scale in; 25 SMRs + 25 band peaks out
# bitalloc: SMRs + reservoir in, allocation + reservoir out; quantize: lines, scale, band peaks, allocation in,
# mantissas + scale factors out
BYTES_ALLOC = (25 * 8 + 4 + 100 + 4) + (8192 + 4 + 25 * 8 + 100 + 4096 + 100)
BYTES_PATH = 12496                                # hop in + all integer outputs


def measured_traffic():
    """HBM bytes per frame per kernel from the committed PMC summary (profiles/*_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 correction applied by
    tools/summarize_profiles.py).  bench.py cannot collect PMC counters itself; {} when no summary exists."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")),
                   key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))])
    if not files:
        return {}, None
    with open(files[-1]) as f:
        d = json.load(f)
    return {k: v["hbm_bytes_per_frame"] for k, v in d["kernels"].items()}, os.path.basename(files[-1])


def measured_valu():
    """VALU instructions per frame and issue-capacity fraction of smr_kernel from the committed SQ counter summary
    (profiles/*_sq_counters.json, tools/summarize_sq.py); None when there is none."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")),
                   key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))])
    if not files:
        return None
    with open(files[-1]) as f:
        k = json.load(f)["kernels"].get("smr_kernel", {})
    if "valu_insts_per_frame" not in k:
        return None
    return {"valu_insts_per_frame": k["valu_insts_per_frame"], "issue_frac_of_peak": k.get("valu_issue_frac_at_4cyc"),
            "source": os.path.basename(files[-1]),
            "note": "wave64 fp64/int VALU instructions (4 cycles each on a SIMD, 1024 SIMDs): the roofline that binds this kernel"}


def make_noise_stream(torch, device, n_frames, seed):
    """C2 content generated on the device: 16-bit Gaussian PCM mapped to signed fractions
    (pcmfile.py:91-100), one leading hop of zeros (priorBlock at file start)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pcm = torch.randn((n_frames * HOP,), generator=g, device=device, dtype=torch.float64) * (0.1 * 32767)
    pcm = torch.clamp(torch.round(pcm), -32767, 32767)
    x = torch.sign(pcm) * 2.0 * torch.abs(pcm) / 65535
    return torch.cat([torch.zeros(HOP, device=device, dtype=torch.float64), x]).contiguous()


def cpu_baseline(n_frames):
    """The faithful one-block-at-a-time NumPy port (oracle.codec), single core, on the same kind of stream."""
    import numpy as np
    from mrcaudiocodec_amd import synth
    from oracle import codec as ocodec, fast
    x = synth.c2_noise(n_frames + 1)
    cp = ocodec.default_params()
    ocodec.EncodeSingleChannel(x[0:2048].copy(), cp)              # warm-up frame
    t0 = time.perf_counter()
    for i in range(1, n_frames + 1):
        cp.bitReservoir = 0
        ocodec.EncodeSingleChannel(x[i * HOP:i * HOP + 2048].copy(), cp)
    dt = time.perf_counter() - t0
    blocks = np.array(fast.blocks_from_stream(x, HOP))
    t1 = time.perf_counter()
    fast.encode_mono_batch(blocks, 1024, 1024)
    dtv = time.perf_counter() - t1
    return {"value": n_frames * HOP / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d long mono frames of the same synthetic noise, oracle.codec.EncodeSingleChannel "
                      "(faithful NumPy port of codecThem.py:281-354), %.1f s" % (n_frames, dt),
            "vectorised_port_value": blocks.shape[0] * HOP / dtv / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1 << 17, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames of the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from mrcaudiocodec_amd.batch import StreamEncoder
    from mrcaudiocodec_amd.shard import shard_frames, max_over_ranks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # MRC_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (several ranks share a card,
    # barrier and max-over-ranks go over gloo on the CPU); the driver's runs use the default, RCCL ("nccl").
    backend = os.environ.get("MRC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    enc = StreamEncoder(device_id=local)
    F = args.frames
    # weak scaling: the job is a world*F-frame batch cut into contiguous per-rank ranges (shard.py); the
    # content is synthetic, so each rank generates its own range (one-hop halo included) instead of receiving it
    first, count = shard_frames(world * F, world, rank)
    assert count == F
    pcm = make_noise_stream(torch, device, F, seed=1234 + rank)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        enc.encode_long(pcm, None, F)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc.encode_long(pcm, None, F)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = max_over_ranks(elapsed, coll_device)

    # per-kernel device time, hipEvents on the launch stream (outside the timed region)
    enc.h.set_timing(True)
    import numpy as np
    reps = max(3, min(args.steps, 10))
    acc = np.zeros(3)
    for _ in range(reps):
        enc.encode_long(pcm, None, F)
        acc += enc.h.stage_ms()
    enc.h.set_timing(False)
    stage_ms = acc / reps

    if rank == 0:
        total_samples = float(F) * HOP * world * args.steps
        names = ["mdct_long_kernel", "smr_kernel", "bitalloc+quantize kernels"]
        per_unit = [BYTES_MDCT, BYTES_SMR, BYTES_ALLOC]
        kernels = []
        for nm, ms, bpu in zip(names, stage_ms, per_unit):
            gbs = bpu * F / (ms * 1e-3) / 1e9
            kernels.append({"name": nm, "ms": round(float(ms), 4), "algorithmic_bytes": bpu * F,
                            "achieved_GBs": round(gbs, 2), "frac_hbm": round(gbs / HBM_PEAK_GBS, 5)})
        dom = int(np.argmax(stage_ms))
        traffic, traffic_src = measured_traffic()
        stage_kernels = [["mdct_long_kernel"], ["smr_kernel"], ["bitalloc_kernel", "quantize_kernel"]]
        for kinfo, parts in zip(kernels, stage_kernels):
            kinfo["traffic"] = round(sum(traffic[p] for p in parts) * F) if all(p in traffic for p in parts) else None
        line = {
            "metric": "encode Msamples/sec (48 kHz, 2048-pt MDCT)",
            "value": round(total_samples / elapsed / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: mono 48 kHz white noise (sigma 0.1 FS, 16-bit grid), 2048-pt long blocks, "
                                   "whole encode path, independent-frames mode",
                       "frames_per_gpu_per_step": F, "hop": HOP, "layout": "hop-overlapped f64 stream in HBM",
                       "parallelism": "frame-sharded x%d, no collective" % world},
            "roofline": {"kernel": names[dom], "bound": "hbm", "achieved": kernels[dom]["achieved_GBs"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kernels[dom]["frac_hbm"],
                         "traffic": kernels[dom]["traffic"], "traffic_source": traffic_src,
                         "limiter": "fp64 VALU issue (masker spreading + FFT + SPL conversions), not HBM -- DESIGN.md section 4",
                         "valu": measured_valu(),
                         "note": "dominant kernel by device time, priced against HBM as the contract asks; the "
                                 "HBM-bound kernel of the path is mdct_long_kernel, see kernels[0]"},
            "kernels": kernels,
            "whole_path": {"algorithmic_bytes_per_frame": BYTES_PATH,
                           "achieved_GBs": round(BYTES_PATH * F * world * args.steps / elapsed / 1e9, 2)},
        }
        if world == 1 and args.cpu_frames > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames)
            line["speedup_vs_cpu_port"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
