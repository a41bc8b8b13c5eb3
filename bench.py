#!/usr/bin/env python3
"""
bench.py -- encode throughput of the MI355X hot path (BASELINE.json metric: encode Msamples/s,
48 kHz, 2048-point MDCT blocks).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (N > 1, one rank per GPU)

A step = one pass of the whole hot path (window+MDCT -> masked threshold/SMR -> bit allocation ->
scale factors + mantissas) over one batch of synthetic PCM frames that is already resident in HBM.

`value` (every N): BASELINE.json configs[1] -- mono 48 kHz Gaussian white noise (sigma 0.1 FS, 16-bit PCM), all-long
blocks (a = b = 1024), independent-frames mode (reservoir_in = 0), F frames per GPU per step.  The content is ONE
stream of world * F frames defined by a counter-based generator (a function of the global sample index); rank r
encodes the contiguous frame range shard_frames() gives it, reading its slice with the one-hop halo -- no collective
on the data path ("weak" scaling: per-GPU work fixed).  The resident layout is the file's own: int16 PCM codes in,
converted on load as pcmfile.py:91-100 does, 16-bit mantissa codes out (`f64_layout` reports the float64-in /
int32-out layout round 1 measured).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant kernel (largest share of device time): algorithmic bytes per launch / its average duration
                (hipEvents on the launch stream) against 8 TB/s HBM; fp64 figures from the committed SQ counters.
  kernels       the same for every kernel of the path.
  host_to_host  N = 1: PCM in page-locked host memory -> codes in page-locked host memory (H2D + kernels + D2H,
                mrc_encode_stream_pcm16, chunks pipelined over 3 HIP streams), median of >= 5 runs.
  configs       N = 1: configs[2] (stereo, joint M/S path) and configs[3] (block switching long/short) resident
                rates with their own roofline objects.
  configs4      N > 1: BASELINE.json configs[4] -- the C3 stereo content as ONE stream of world * F' frames, frame-
                sharded with halo, joint path; whole-job Msamples/s over the max-over-ranks time, and the host
                Huffman + bit-packing rate of the ranks' outputs (`host_pack_Msamples_s`, reported separately).
  cpu_baseline  N = 1: the oracle's faithful NumPy port of the reference path on the host: 1 core, and all cores
                (one process per core over disjoint frame ranges, core count stated).
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
NB = 25                         # scale-factor bands of a long block at 48 kHz
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6         # vector fp64 (= matrix fp64 on MI355X)


# ---------------------------------------------------------------------------------------------- algorithmic bytes
def algorithmic_bytes(joint, pcm16, mant16, a=1024, b=1024, nb=NB):
    """Per block and kernel.  Long mono block, f64 layout: the figures of SURVEY.md 8(d)."""
    half = (a + b) // 2
    smp = 2 if pcm16 else 8
    nsig, nch, nstream = (4, 2, 2) if joint else (1, 1, 1)
    new = b * smp * nch                                        # every hop counted once, although two frames read it
    lines = half * 8
    mant = half * (2 if mant16 else 4)
    return {
        "mdct": new + nsig * lines,
        "smr": new + nsig * (lines + 4 + 2 * nb * 8),           # lines + overall scale in; SMRs + band peaks out
        "band_stats": (2 * lines + 4 * nb) if joint else 0,     # L, R lines in; M/S switch out
        "bitalloc": nsig * nb * 8 + 4 + (4 * nb if joint else 0) + nstream * 4 * nb + 4,
        "quantize": nstream * (lines + mant + 4 * nb * 3) + nsig * (4 + 8 * nb),
        "path": new + nstream * (mant + 8 * nb) + 4 * nsig + (4 * nb if joint else 0) + 4,
    }


KERNEL_NAMES = ["mdct", "smr", "band_stats", "bitalloc", "quantize"]
KERNEL_LABEL = {"mdct": "mdct_long_kernel", "smr": "smr_kernel", "band_stats": "ms_switch_kernel",
                "bitalloc": "bitalloc_kernel", "quantize": "quantize_kernel"}


def latest_profile(suffix, tag=None):
    import glob
    import re
    pat = "*%s" % suffix if tag is None else "*%s*%s" % (tag, suffix)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)),
                   key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))])
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def measured_traffic(tag=None):
    """HBM bytes per frame per kernel from the committed PMC summary (profiles/*_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 correction applied by tools/summarize_profiles.py).
    bench.py cannot collect PMC counters itself; {} when no summary exists."""
    d, name = latest_profile("_traffic.json", tag)
    if not d:
        return {}, None
    return {k: v["hbm_bytes_per_frame"] for k, v in d["kernels"].items()}, name


def measured_valu(smr_ms, frames):
    """fp64 work of smr_kernel from the committed SQ counter summary (profiles/*_sq_counters.json): VALU instructions per
    frame, the fp64 share of them, and the fp64 rate they amount to at the launch duration measured in THIS run."""
    d, name = latest_profile("_sq_counters.json")
    if not d:
        return None
    k = d["kernels"].get("smr_kernel", {})
    if "valu_insts_per_frame" not in k:
        return None
    fr = float(d.get("frames_per_launch", 1))
    out = {"valu_insts_per_frame": k["valu_insts_per_frame"], "issue_frac_of_peak": k.get("valu_issue_frac_at_4cyc"),
           "valu_busy_frac": k.get("valu_busy_frac"), "source": name}
    if all(c in k for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU")):
        f64_insts = k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_FMA_F64"]
        flops_per_frame = (k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + 2 * k["SQ_INSTS_VALU_FMA_F64"]) * 64 / fr
        tf = flops_per_frame * frames / (smr_ms * 1e-3) / 1e12
        out.update({"fp64_share_of_valu_insts": round(f64_insts / k["SQ_INSTS_VALU"], 3),
                    "fp64_flops_per_frame": round(flops_per_frame), "fp64_TFLOPs": round(tf, 2),
                    "fp64_frac_of_peak": round(tf / FP64_PEAK_TFLOPS, 4), "fp64_peak_TFLOPs": FP64_PEAK_TFLOPS})
    if "SQ_LDS_BANK_CONFLICT" in k and "SQ_LDS_IDX_ACTIVE" in k:
        out["lds_bank_conflict_share"] = round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"], 3)
    out["note"] = "wave64 fp64/int VALU instructions (4 cycles each on a SIMD, 1024 SIMDs): the roofline that binds this kernel"
    return out


# ---------------------------------------------------------------------------------------------- synthetic content
def _mix(torch, x):
    """splitmix64 finaliser on int64 tensors (multiplication wraps; logical right shifts emulated by masking)."""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    x = (x ^ lsr(x, 30)) * -4658895280553007687          # 0xBF58476D1CE4E5B9
    x = (x ^ lsr(x, 27)) * -7723592293110705685          # 0x94D049BB133111EB
    return x ^ lsr(x, 31)


def gauss_at(torch, idx, seed):
    """Standard normal values as a FUNCTION of the global sample index (counter-based: every rank can produce its own
    slice of ONE stream without communication): two hashed uniforms -> Box-Muller."""
    k = idx * 2 + (((seed * 0x9E3779B97F4A7C15) + (1 << 63)) % (1 << 64) - (1 << 63))     # wrapped to int64
    u1 = ((_mix(torch, k) >> 11) & ((1 << 53) - 1)).to(torch.float64) * 2.0 ** -53
    u2 = ((_mix(torch, k + 1) >> 11) & ((1 << 53) - 1)).to(torch.float64) * 2.0 ** -53
    return torch.sqrt(-2.0 * torch.log(1.0 - u1)) * torch.cos(6.283185307179586 * u2)


def to_pcm16(torch, v):
    return torch.clamp(torch.round(v), -32767, 32767).to(torch.int16)


def stream_slice(torch, device, kind, first_frame, n_frames):
    """int16 PCM of frames first_frame .. first_frame + n_frames - 1 of the global stream, one-hop halo in front:
    stream positions [first_frame * HOP, (first_frame + n_frames + 1) * HOP); position p holds global sample p - HOP
    (the leading hop is the zero priorBlock of the file start).  kind: 'c2' mono noise (sigma 0.1), 'c3' stereo
    (L = g1; R = 0.8 g1 + 0.2 g2 on even hops, 0.1 g2 on odd hops), 'c4' noise floor sigma 0.01 with a sigma 0.5 burst
    of 128 samples at the start of every 5th hop.  -> list of int16 tensors (one per channel)."""
    pos = torch.arange(first_frame * HOP, (first_frame + n_frames + 1) * HOP, device=device, dtype=torch.int64)
    s = pos - HOP
    live = s >= 0
    if kind == "c2":
        return [torch.where(live, to_pcm16(torch, gauss_at(torch, s, 1234) * (0.1 * 32767)), 0).contiguous()]
    if kind == "c3":
        g1 = to_pcm16(torch, gauss_at(torch, s, 1234) * (0.1 * 32767)).to(torch.float64)
        g2 = to_pcm16(torch, gauss_at(torch, s, 5678) * (0.1 * 32767)).to(torch.float64)
        even = (torch.div(s, HOP, rounding_mode="floor") % 2) == 0
        r = to_pcm16(torch, torch.where(even, 0.8 * g1 + 0.2 * g2, 0.1 * g2))
        z = torch.zeros((), dtype=torch.int16, device=device)
        return [torch.where(live, g1.to(torch.int16), z).contiguous(), torch.where(live, r, z).contiguous()]
    if kind == "c4":
        hop = torch.div(s, HOP, rounding_mode="floor")
        burst = ((hop % 5) == 4) & ((s - hop * HOP) < 128)
        sigma = torch.where(burst, 0.5 * 32767, 0.01 * 32767)
        return [torch.where(live, to_pcm16(torch, gauss_at(torch, s, 42) * sigma), 0).contiguous()]
    raise ValueError(kind)


def c4_shapes(n_hops):
    """The forced cycle of pacfileThem.py:1192-1210 around every 5th (burst) hop: -> {(a, b): [offsets]}."""
    by_shape = {}
    off, a = 0, HOP
    for h in range(n_hops):
        if h % 5 == 4:
            for _ in range(8):
                by_shape.setdefault((a, 128), []).append(off); off += a; a = 128
        else:
            by_shape.setdefault((a, HOP), []).append(off); off += a; a = HOP
    return by_shape


# ---------------------------------------------------------------------------------------------- CPU baseline
def _c2_noise(np, n_frames, seed=1234, sigma=0.1):
    """C2 content on the host (same recipe as mrcaudiocodec_amd.synth.c2_noise; restated so that the worker processes
    of the all-cores baseline import nothing that loads the HIP runtime)."""
    pcm = np.clip(np.rint(np.random.default_rng(seed).normal(0.0, sigma * 32767, n_frames * HOP)), -32767, 32767)
    return np.concatenate([np.zeros(HOP), np.sign(pcm) * 2.0 * np.abs(pcm) / 65535])


def _cpu_range(args):
    first, n = args
    import numpy as np
    from oracle import codec as ocodec
    x = _c2_noise(np, first + n + 1)
    cp = ocodec.default_params()
    t0 = time.perf_counter()
    for i in range(first, first + n):
        cp.bitReservoir = 0
        ocodec.EncodeSingleChannel(x[i * HOP:i * HOP + 2048].copy(), cp)
    return time.perf_counter() - t0


def cpu_baseline(n_frames):
    """The faithful one-block-at-a-time NumPy port (oracle.codec), on the same kind of stream: 1 core, then one
    process per host core over disjoint frame ranges."""
    import multiprocessing as mp
    import numpy as np
    from oracle import fast
    _cpu_range((0, 1))                                           # warm-up frame
    dt = _cpu_range((1, n_frames))
    x = _c2_noise(np, n_frames + 1)
    blocks = np.array(fast.blocks_from_stream(x, HOP))
    t1 = time.perf_counter()
    fast.encode_mono_batch(blocks, 1024, 1024)
    dtv = time.perf_counter() - t1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                               # one GPU's share of the host
    per = max(8, -(-n_frames // 2))                              # ~ half the single-core sample per worker
    t2 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_cpu_range, [(1 + i * per, per) for i in range(cores)])
    wall = time.perf_counter() - t2
    return {"value": n_frames * HOP / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d long mono frames of the same kind of synthetic noise, oracle.codec.EncodeSingleChannel "
                      "(faithful NumPy port of codecThem.py:281-354), %.1f s" % (n_frames, dt),
            "vectorised_port_value": blocks.shape[0] * HOP / dtv / 1e6,
            "all_cores": {"value": cores * per * HOP / wall / 1e6, "unit": "Msamples/s", "cores": cores,
                          "sample": "%d processes x %d frames over disjoint frame ranges, wall %.1f s (process start-up "
                                    "included)" % (cores, per, wall)}}


# ---------------------------------------------------------------------------------------------- measurement helpers
def kernel_report(names_ms, bytes_per_unit, units, traffic, label=KERNEL_LABEL):
    rows = []
    for nm, ms in names_ms:
        if ms <= 0 or bytes_per_unit.get(nm, 0) == 0:
            continue
        gbs = bytes_per_unit[nm] * units / (ms * 1e-3) / 1e9
        t = traffic.get(label[nm])
        rows.append({"name": label[nm], "ms": round(float(ms), 4), "algorithmic_bytes": int(bytes_per_unit[nm] * units),
                     "achieved_GBs": round(gbs, 2), "frac_hbm": round(gbs / HBM_PEAK_GBS, 5),
                     "traffic": None if t is None else round(t * units)})
    return rows


def roofline_of(rows, extra=None):
    dom = max(rows, key=lambda r: r["ms"])
    out = {"kernel": dom["name"], "bound": "hbm", "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": dom["frac_hbm"], "traffic": dom["traffic"]}
    if extra:
        out.update(extra)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1 << 17, help="mono frames per GPU per step")
    ap.add_argument("--h2h-frames", type=int, default=0, help="frames of the host-to-host stream (0 = 4 x --frames)")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--skip-extras", action="store_true", help="only the headline measurement (profiling runs)")
    ap.add_argument("--only", choices=["stereo", "switch"], default=None,
                    help="profiling runs: ONLY the timed loop of configs[2] / configs[3] (prints a short JSON line)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_line = None
    if world == 1 and args.cpu_frames > 0 and not args.skip_extras:
        # first of all, before this process touches the GPU: the baseline starts one worker process per core
        for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ.setdefault(v, "1")
        cpu_line = cpu_baseline(args.cpu_frames)
    import numpy as np
    import torch
    import torch.distributed as dist
    from mrcaudiocodec_amd import PinnedArray, pacfile as ppac
    from mrcaudiocodec_amd.batch import StreamEncoder
    from mrcaudiocodec_amd.shard import shard_frames, max_over_ranks

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

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed_steps(fn):
        """W warm-up steps, then exactly K steps between barriers; max over ranks of the elapsed time."""
        for _ in range(args.warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        barrier()
        return max_over_ranks(elapsed, coll_device)

    def kernel_ms(fn, reps=None):
        """per-kernel device time (hipEvents on the launch stream), outside the timed region; fn returns after ONE encode"""
        enc.h.set_timing(True)
        reps = reps or max(3, min(args.steps, 10))
        acc = np.zeros(5)
        for _ in range(reps):
            fn()
            acc += enc.h.kernel_ms()
        enc.h.set_timing(False)
        return acc / reps

    if args.only:
        # a profiler is watching: run nothing but this configuration's kernels
        if args.only == "stereo":
            Fs = F // 2
            sl, sr = stream_slice(torch, device, "c3", 0, Fs)
            el = timed_steps(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
            print(json.dumps({"only": "stereo", "frames": Fs, "ms_per_step": el / args.steps * 1e3,
                              "Msamples_s": 2.0 * Fs * HOP * args.steps / el / 1e6}))
        else:
            (xs,) = stream_slice(torch, device, "c4", 0, F)
            groups = {k: torch.tensor(v, dtype=torch.int64, device=device) for k, v in c4_shapes(F).items()}

            def run_switched():
                for (a, b), o in groups.items():
                    enc.encode(a, b, xs, None, o.numel(), 0, o, mantissa16=True, offsets_checked=True)
            el = timed_steps(run_switched)
            print(json.dumps({"only": "switch", "hops": F, "ms_per_step": el / args.steps * 1e3,
                              "Msamples_s": float(F) * HOP * args.steps / el / 1e6}))
        return

    # ---- headline: configs[1], one global stream, rank r takes frames [first, first + F)
    first, count = shard_frames(world * F, world, rank)
    assert count == F
    (pcm,) = stream_slice(torch, device, "c2", first, F)
    elapsed = timed_steps(lambda: enc.encode_long(pcm, None, F, mantissa16=True))
    kms = kernel_ms(lambda: enc.encode_long(pcm, None, F, mantissa16=True))
    line = None
    if rank == 0:
        total_samples = float(F) * HOP * world * args.steps
        ab = algorithmic_bytes(False, True, True)
        traffic, traffic_src = measured_traffic("mono")
        rows = kernel_report(list(zip(KERNEL_NAMES, kms)), ab, F, traffic)
        smr_ms = float(kms[1])
        line = {
            "metric": "encode Msamples/sec (48 kHz, 2048-pt MDCT)",
            "value": round(total_samples / elapsed / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: mono 48 kHz white noise (sigma 0.1 FS, 16-bit PCM), 2048-pt long blocks, "
                                   "whole encode path, independent-frames mode",
                       "frames_per_gpu_per_step": F, "hop": HOP,
                       "layout": "resident in HBM: int16 PCM codes in (hop-overlapped stream, converted on load as "
                                 "pcmfile.py:91-100), uint16 mantissa plane + int32 side info out",
                       "parallelism": "one global stream of %d frames, frame-sharded x%d with a one-hop halo, no collective"
                                      % (world * F, world)},
            "roofline": roofline_of(rows, {
                "traffic_source": traffic_src,
                "limiter": "fp64 VALU issue (masker spreading + FFT + SPL conversions), not HBM -- DESIGN.md section 4",
                "valu": measured_valu(smr_ms, F),
                "note": "dominant kernel by device time, priced against HBM as the contract asks; the HBM-bound kernel "
                        "of the path is mdct_long_kernel, see kernels[0]"}),
            "kernels": rows,
            "whole_path": {"algorithmic_bytes_per_frame": ab["path"],
                           "achieved_GBs": round(ab["path"] * F * world * args.steps / elapsed / 1e9, 2)},
        }

    if not args.skip_extras:
        # ---- the float64-in / int32-out layout round 1 measured (continuity)
        x64 = (torch.sign(pcm.double()) * 2.0 * torch.abs(pcm.double()) / 65535).contiguous()
        e64 = timed_steps(lambda: enc.encode_long(x64, None, F))
        k64 = kernel_ms(lambda: enc.encode_long(x64, None, F))
        if rank == 0:
            ab64 = algorithmic_bytes(False, False, False)
            line["f64_layout"] = {"value": round(float(F) * HOP * world * args.steps / e64 / 1e6, 3), "unit": "Msamples/s",
                                  "ms_per_step": round(e64 / args.steps * 1e3, 4),
                                  "layout": "float64 signed fractions in, int32 mantissa plane out (round-1 layout)",
                                  "kernels": kernel_report(list(zip(KERNEL_NAMES, k64)), ab64, F, {})}
        del x64

    if world == 1 and not args.skip_extras:
        # ---- SURVEY.md 8(d) metric as defined: page-locked host PCM -> codes in page-locked host memory
        keep = []

        def pin(shape, dt):
            p = PinnedArray(shape, dt); keep.append(p); return p.array
        # a stream of 4 F frames (SURVEY 8(d) C2 is 2^20): the pipeline's fill and drain (one chunk each) weigh less
        Fh = args.h2h_frames or 4 * F
        host_pcm = pin(((Fh + 1) * HOP,), np.int16)
        for f0 in range(0, Fh, F):                          # generated on the device slice by slice (counter-based)
            nfr = min(F, Fh - f0)
            (part,) = stream_slice(torch, device, "c2", f0, nfr)
            host_pcm[f0 * HOP:(f0 + nfr + 1) * HOP] = part.cpu().numpy()
            del part
        outs = dict(overall_scale=pin((Fh, 1), np.int32), scale_factor=pin((Fh, 1, NB), np.int32),
                    bit_alloc=pin((Fh, 1, NB), np.int32), mantissa=pin((Fh, 1, HOP), np.uint16),
                    reservoir_out=pin((Fh,), np.int32))
        runs = {}
        for chunk in (16384, 32768, 65536):
            enc.h.encode_stream_pcm16(host_pcm, None, None, chunk, outs)        # warm-up: lane buffers, first touch
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                enc.h.encode_stream_pcm16(host_pcm, None, None, chunk, outs)
                ts.append(time.perf_counter() - t0)
            runs[chunk] = float(np.median(ts))
        best = min(runs, key=runs.get)
        nchk = min(F, Fh)
        dev_out = enc.encode_long(pcm[:(nchk + 1) * HOP].contiguous(), None, nchk, mantissa16=True)
        same = bool(np.array_equal(dev_out["mantissa"].cpu().numpy().view(np.uint16), outs["mantissa"][:nchk]))
        pcie = 2 * HOP + 2 * HOP + 2 * 4 * NB + 8
        line["host_to_host"] = {
            "value": round(Fh * HOP / runs[best] / 1e6, 3), "unit": "Msamples/s", "frames": Fh, "median_of": 5,
            "chunk_frames": best, "by_chunk_frames": {str(k): round(Fh * HOP / v / 1e6, 1) for k, v in runs.items()},
            "what": "int16 PCM in page-locked host memory -> H2D -> kernels -> D2H -> uint16 mantissas + int32 side info "
                    "in page-locked host memory (mrc_encode_stream_pcm16, 3 HIP streams)",
            "pcie_bytes_per_frame": pcie, "pcie_GBs_each_way": round(Fh * 2 * HOP / runs[best] / 1e9, 2),
            "pcie_ceiling": "page-locked copies on this box (tools/pcie_rates.py): 55-57 GB/s one way alone, 25 / 47 GB/s "
                            "each way with both directions busy at 16 / 64 MiB per copy",
            "equals_resident_result": same}
        del dev_out
        # ... and with the back end on the device as well: the same PCM -> `.pac` chunk bytes in page-locked host memory
        pac_buf = pin((Fh * HOP + 4096,), np.uint8)
        pac_chunk = best
        enc.h.encode_stream_pcm16_pac(host_pcm, None, None, True, pac_chunk, {"bytes": pac_buf})
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            pac = enc.h.encode_stream_pcm16_pac(host_pcm, None, None, True, pac_chunk, {"bytes": pac_buf})
            ts.append(time.perf_counter() - t0)
        tp = float(np.median(ts))
        line["host_to_host"]["pac"] = {
            "value": round(Fh * HOP / tp / 1e6, 3), "unit": "Msamples/s", "chunk_frames": pac_chunk,
            "bytes_per_frame": round(pac["bytes"].size / Fh, 1),
            "what": "the same PCM -> .pac chunk bytes (Huffman pricing + bit packing on the device, mrc_encode_stream_pcm16_pac) "
                    "in page-locked host memory"}
        del pac
        for p in keep:
            p.free()

        # ---- configs[2]: stereo, joint M/S path, resident
        Fs = F // 2
        sl, sr = stream_slice(torch, device, "c3", 0, Fs)
        es = timed_steps(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        ks = kernel_ms(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        out = enc.encode_long(sl, sr, Fs, mantissa16=True)
        abj = algorithmic_bytes(True, True, True)
        tj, tj_src = measured_traffic("joint")
        rows_j = kernel_report(list(zip(KERNEL_NAMES, ks)), abj, Fs, tj)
        cfgs = {"stereo_ms": {
            "workload": "configs[2]: stereo 48 kHz (C3 content), joint path with the M/S decision, long blocks",
            "value": round(2.0 * Fs * HOP * args.steps / es / 1e6, 3), "unit": "Msamples/s", "frames": Fs,
            "ms_per_step": round(es / args.steps * 1e3, 4),
            "ms_switch_on_fraction": round(float(out["ms_switch"].double().mean().item()), 3),
            "roofline": roofline_of(rows_j, {"traffic_source": tj_src}), "kernels": rows_j}}
        # host back end on these outputs (reported separately)
        cfgs["stereo_ms"]["host_pack"] = host_pack_rate(np, ppac, out, min(Fs, 16384))
        cfgs["stereo_ms"]["device_pack"] = device_pack_rate(torch, enc, out, Fs, args.steps)
        del sl, sr, out

        # ---- configs[3]: block switching (long / start / 8 short / stop), mono, resident
        hops = F
        (xs,) = stream_slice(torch, device, "c4", 0, hops)
        groups = {k: torch.tensor(v, dtype=torch.int64, device=device) for k, v in c4_shapes(hops).items()}

        def run_switched():
            for (a, b), o in groups.items():
                enc.encode(a, b, xs, None, o.numel(), 0, o, mantissa16=True, offsets_checked=True)
        eb = timed_steps(run_switched)
        per_shape, rows_b = {}, []
        tb, tb_src = measured_traffic("switch")
        for (a, b), o in groups.items():
            km = kernel_ms(lambda: enc.encode(a, b, xs, None, o.numel(), 0, o, mantissa16=True, offsets_checked=True))
            nb = len(enc.h.bands(a, b))
            abk = algorithmic_bytes(False, True, True, a, b, nb)
            label = dict(KERNEL_LABEL)
            if (a, b) != (HOP, HOP):
                label["mdct"] = "mdct_kernel"
            rr = kernel_report(list(zip(KERNEL_NAMES, km)), abk, o.numel(), {}, label)   # (PMC traffic: per config, below)
            for r in rr:
                r["shape"] = "%dx%d" % (a, b)
            rows_b += rr
            per_shape["%dx%d" % (a, b)] = {"blocks": int(o.numel()), "kernel_ms": [round(float(v), 4) for v in km]}
        cfgs["block_switching"] = {
            "workload": "configs[3]: mono stream with a burst every 5th hop -> (1024,1024), (1024,128), 7x(128,128), (128,1024) "
                        "blocks, one launch set per shape",
            "value": round(float(hops) * HOP * args.steps / eb / 1e6, 3), "unit": "Msamples/s", "hops": hops,
            "ms_per_step": round(eb / args.steps * 1e3, 4), "per_shape": per_shape,
            "roofline": roofline_of(rows_b, {"traffic_source": tb_src,
                                             "traffic": None if "smr_kernel" not in tb else round(tb["smr_kernel"] * hops),
                                             "traffic_note": "smr_kernel, all four block shapes of a step together"}),
            "hbm_traffic_bytes_per_hop": {k: v for k, v in tb.items()} or None, "kernels": rows_b}
        # ---- stream mode: MANY stereo streams advance one block per step, every stream's bit reservoir chained through the
        # Huffman savings of its previous block on the device (codecThem.py:224,274) -- the mode that writes the files the
        # reference writes; a step's batch is the number of streams, not the length of one
        nS, nT = 8192, 12
        gs = torch.Generator(device=device)
        gs.manual_seed(7)
        pl = torch.clamp(torch.round(torch.randn((nS, (nT + 1) * HOP), generator=gs, device=device, dtype=torch.float64) * 3000),
                         -32767, 32767)
        ssl = (torch.sign(pl) * 2.0 * torch.abs(pl) / 65535).contiguous()
        ssl[:, :HOP] = 0
        ssr = (0.7 * ssl + 0.3 * torch.roll(ssl, 17, dims=1)).contiguous()
        ssr[:, :HOP] = 0
        del pl
        one = np.array([(i * HOP, HOP, HOP) for i in range(nT)], dtype=np.int64)
        shapes_all = np.broadcast_to(one, (nS, nT, 3))
        # warm-up with the WHOLE schedule: every step keeps its outputs (the packer reads them afterwards), 70 MB each --
        # the caching allocator then holds the blocks the timed run takes again
        warm = enc.encode_chained(ssl, ssr, shapes_all)
        del warm
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        _, reservoir = enc.encode_chained(ssl, ssr, shapes_all)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        cfgs["stream_mode"] = {
            "workload": "%d stereo streams x %d chained joint long blocks, bit reservoirs carried from block to block on the "
                        "device (encode kernels + Huffman pricing per step)" % (nS, nT),
            "value": round(2.0 * nS * nT * HOP / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3 / nT, 4),
            "mean_final_reservoir_bits": round(float(reservoir.double().mean().item()), 1)}
        del ssl, ssr
        line["configs"] = cfgs
        del xs, groups

    if world > 1 and not args.skip_extras:
        # ---- configs[4]: the C3 stereo stream, frame-sharded, joint path, host pack reported separately
        Fs = F // 2
        first_s, cnt = shard_frames(world * Fs, world, rank)
        sl, sr = stream_slice(torch, device, "c3", first_s, Fs)
        es = timed_steps(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        out = enc.encode_long(sl, sr, Fs, mantissa16=True)
        pack = host_pack_rate(np, ppac, out, min(Fs, 16384), threads=max(1, min(16, (os.cpu_count() or 8) // world)))
        pack_rate = pack["huffman_priced_on_host_Msamples_s"]
        slowest = -max_over_ranks(-pack_rate, coll_device)            # min over ranks
        dpack = device_pack_rate(torch, enc, out, Fs, args.steps)
        dslow = -max_over_ranks(-dpack["huffman_priced_on_device_Msamples_s"], coll_device)
        if rank == 0:
            line["configs4"] = {
                "workload": "configs[4]: C3 stereo content as ONE stream of %d frames, frame-sharded x%d (contiguous ranges, "
                            "one-hop halo), joint path, independent-frames mode" % (world * Fs, world),
                "value": round(2.0 * Fs * HOP * world * args.steps / es / 1e6, 3), "unit": "Msamples/s",
                "frames_per_gpu_per_step": Fs, "ms_per_step": round(es / args.steps * 1e3, 4),
                "host_pack_Msamples_s": round(slowest * world, 1),
                "device_pack_Msamples_s": round(dslow * world, 1),
                "device_pack": dict(dpack, note="the same back end on each rank's GPU (mrc_dev_pack_blocks); whole-job figure = "
                                                "world x the slowest rank's rate"),
                "host_pack": dict(pack, note="Huffman table choice + bit packing of each rank's outputs on its share of the "
                                             "host cores (C++, csrc/mrc_pack.cpp), outside the timed GPU region; whole-job "
                                             "figure = world x the slowest rank's rate")}

    if rank == 0:
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
            line["speedup_vs_cpu_port"] = round(line["value"] / cpu_line["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def device_pack_rate(torch, enc, out, n, steps):
    """The same back end ON THE DEVICE (mrc_dev_pack_blocks: Huffman pricing, chunk sizes, bit packing of the outputs
    where the encoder left them), reported beside the host packer; not part of `value`."""
    L = enc.h.cfg.n_mdct_lines
    res = {}
    for name, use in (("huffman_priced_on_device", True), ("raw", False)):
        packed = enc.pack(L, L, out, use_huffman=use)               # warm-up, buffer sizing
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            packed = enc.pack(L, L, out, use_huffman=use)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name + "_Msamples_s"] = round(2.0 * n * L / dt / 1e6, 1)
        res[name + "_bytes_per_frame"] = round(packed["bytes"].numel() / n, 1)
    res["note"] = ("joint chunks of %d stereo frames, resident in HBM -> .pac bytes in HBM; one synchronisation per call "
                   "(the byte count)" % n)
    return res


def host_pack_rate(np, ppac, out, n, threads=None):
    """C++ packer on the first n joint blocks of a device result: raw, Huffman priced on the host, tables given."""
    cfg = ppac.make_config()
    if threads:
        ppac.set_threads(threads)
    host = {k: out[k][:n].cpu().numpy() for k in ("overall_scale", "ms_switch", "scale_factor", "bit_alloc")}
    mant = out["mantissa"][:n].cpu().numpy().view(np.uint16)        # the 16-bit plane goes to the packer as it is
    args = (cfg, 1024, 1024, host["overall_scale"], host["ms_switch"], host["scale_factor"], host["bit_alloc"], mant)
    res = {"threads": ppac.get_threads(), "blocks": int(n)}
    samples = 2.0 * n * HOP
    _, _, tables, _ = ppac.pack_joint_blocks(*args, use_huffman=True)
    for name, kw in (("raw", dict(use_huffman=False)), ("huffman_priced_on_host", dict(use_huffman=True)),
                     ("huffman_tables_given", dict(huff_table=tables))):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ppac.pack_joint_blocks(*args, **kw); ts.append(time.perf_counter() - t0)
        res[name + "_Msamples_s"] = round(samples / min(ts) / 1e6, 1)
    res["huffman_chunks_frac"] = round(float((tables != 15).mean()), 3)
    return res


if __name__ == "__main__":
    main()
