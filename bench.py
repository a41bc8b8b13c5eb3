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
  configs       N = 1: configs[2] (stereo, joint M/S path) and configs[3] (block switching long/short, block shapes from
                the transient detector on the content, its kernel inside the timed step) resident rates with their own
                roofline objects; stream_mode (many stereo streams, reservoirs chained) and single_stream (ONE long
                stereo stream -> one .pac file) through the chained call (mrc_*_encode_chained_pac).
  configs4      N > 1: BASELINE.json configs[4] -- the C3 stereo content as ONE stream of world * F' frames
                (F' = 10^7 / 8 per GPU: 10^7 frames at N = 8), frame-sharded with halo, joint path; whole-job Msamples/s
                over the max-over-ranks time, the host Huffman + bit-packing rate of the ranks' outputs
                (`host_pack_Msamples_s`, reported separately), a per-rank host_to_host leg (every rank its own page-locked
                buffers, max over ranks: PCIe / NUMA contention shows here) and stream_mode with whole streams per rank.
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
def algorithmic_bytes(joint, pcm16, mant16, a=1024, b=1024, nb=NB, coded_line_frac=1.0):
    """Per block and kernel.  Long mono block, f64 layout: the figures of SURVEY.md 8(d).
    coded_line_frac: the share of lines in bands that were given bits -- quantize_kernel does not load the others
    (their codes are 0 whatever the line holds), so they are not algorithmic bytes of it."""
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
        "quantize": nstream * (coded_line_frac * lines + mant + 4 * nb * 3) + nsig * (4 + 8 * nb),
        "path": new + nstream * (mant + 8 * nb) + 4 * nsig + (4 * nb if joint else 0) + 4,
    }


def coded_line_fraction(np, out, bands):
    """share of lines that lie in bands with a non-zero bit allocation (from an encode result)"""
    ba = out["bit_alloc"].cpu().numpy()
    w = np.asarray(bands, dtype=np.float64)
    return float(((ba > 0) * w).sum() / (ba.shape[0] * ba.shape[1] * w.sum()))


KERNEL_NAMES = ["mdct", "smr", "band_stats", "bitalloc", "quantize"]
KERNEL_LABEL = {"mdct": "mdct_long_kernel", "smr": "smr_kernel", "band_stats": "ms_switch_direct_kernel",
                "bitalloc": "bitalloc_kernel", "quantize": "quantize_kernel"}


def latest_profile(suffix, tag=None):
    import glob
    import re
    pat = "*%s" % suffix if tag is None else "*%s*%s" % (tag, suffix)
    def order(f):
        # newest round first in the key's last place; within a round the "_final_" set wins over mid-round states
        b = os.path.basename(f)
        m = re.match(r"r(\d+)_", b)
        return (int(m.group(1)) if m else -1, "_final_" in b,
                [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", b)])
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), key=order)
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
    if kind == "c5":
        # one long stereo programme: noise floor + a 440 Hz tone, a 128-sample burst every 37th hop (detector food)
        hop = torch.div(s, HOP, rounding_mode="floor")
        burst = ((hop % 37) == 36) & ((s - hop * HOP) < 128)
        g1, g2 = gauss_at(torch, s, 77), gauss_at(torch, s, 78)
        tone = 0.2 * 32767 * torch.sin(s.to(torch.float64) * (2 * 3.141592653589793 * 440.0 / 48000))
        l = torch.where(burst, g1 * (0.5 * 32767), g1 * (0.02 * 32767) + tone)
        r = torch.where(burst, g1 * (0.4 * 32767), (0.7 * g1 + 0.3 * g2) * (0.02 * 32767) + 0.9 * tone)
        z = torch.zeros((), dtype=torch.int16, device=device)
        return [torch.where(live, to_pcm16(torch, l), z).contiguous(), torch.where(live, to_pcm16(torch, r), z).contiguous()]
    raise ValueError(kind)


def stream_slices(torch, device, kind, first_frame, n_frames, piece=1 << 17):
    """stream_slice for long ranges: generated piece by piece into preallocated tensors (the generator's float64
    temporaries are eight times the size of the int16 result)."""
    nch = 1 if kind in ("c2", "c4") else 2
    out = [torch.empty(((n_frames + 1) * HOP,), dtype=torch.int16, device=device) for _ in range(nch)]
    for f0 in range(0, max(n_frames, 1), piece):
        n = min(piece, n_frames - f0)
        part = stream_slice(torch, device, kind, first_frame + f0, n)
        for c in range(nch):
            out[c][f0 * HOP:(f0 + n + 1) * HOP] = part[c]
        del part
    return out


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
    ap.add_argument("--frames", type=int, default=1 << 20, help="mono frames per GPU per step of the headline (SURVEY 8(d) C2: 2^20)")
    ap.add_argument("--extra-frames", type=int, default=1 << 17, help="frames per step of the other single-GPU configurations")
    ap.add_argument("--h2h-frames", type=int, default=1 << 19, help="frames of the host-to-host stream")
    ap.add_argument("--c4-frames", type=int, default=10 ** 7 // 8,
                    help="N > 1: stereo frames per GPU per step of configs[4] (10^7 / 8: 10^7 frames at N = 8)")
    ap.add_argument("--single-hops", type=int, default=1 << 16, help="hops of the single-stream configuration")
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
    (pcm,) = stream_slices(torch, device, "c2", first, F)
    elapsed = timed_steps(lambda: enc.encode_long(pcm, None, F, mantissa16=True))
    kms = kernel_ms(lambda: enc.encode_long(pcm, None, F, mantissa16=True))
    coded = coded_line_fraction(np, enc.encode_long(pcm, None, F, mantissa16=True), enc.h.bands(HOP, HOP))
    line = None
    if rank == 0:
        total_samples = float(F) * HOP * world * args.steps
        ab = algorithmic_bytes(False, True, True, coded_line_frac=coded)
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
            "coded_line_fraction": round(coded, 4),
            "whole_path": {"algorithmic_bytes_per_frame": ab["path"],
                           "achieved_GBs": round(ab["path"] * F * world * args.steps / elapsed / 1e9, 2)},
        }

    Fx = min(args.extra_frames, F)                      # batch of the other configurations
    if not args.skip_extras and world == 1:
        # ---- the sensitivity certificate of one headline step (outside the timed region): how many of its integer decisions
        # lay within a guard band of floating-point rounding (mrc_get_sensitivity) -- what "bit-exact" means at this size
        enc.h.set_option(5, 1)
        enc.h.sensitivity()
        t0 = time.perf_counter()
        enc.encode_long(pcm, None, F, mantissa16=True)
        torch.cuda.synchronize(device)
        t_sens = time.perf_counter() - t0
        cert = enc.h.sensitivity()
        enc.h.set_option(5, 0)
        cert["decisions_near_an_edge"] = sum(cert[k] for k in ("quantiser_edges", "bitalloc_near_ties", "ms_switch_near_threshold",
                                                               "peak_near_ties"))
        cert["ms_with_counting"] = round(t_sens * 1e3, 3)
        cert["guards"] = ("lines 4e-13 of the scaled block peak at a mantissa / scale-factor edge; SMR pairs 1e-9 dB from a multiple "
                          "of 6 dB; M/S test 1e-12 (relative) from its 0.8 threshold; spectral bins 1e-11 (relative) from a "
                          "neighbour they must beat (include/mrc_hip.h, MRC_SENS_*)")
        line["sensitivity"] = cert
    if not args.skip_extras:
        # ---- the float64-in / int32-out layout round 1 measured (continuity)
        p64 = pcm[:(Fx + 1) * HOP].double()
        x64 = (torch.sign(p64) * 2.0 * torch.abs(p64) / 65535).contiguous()
        del p64
        e64 = timed_steps(lambda: enc.encode_long(x64, None, Fx))
        k64 = kernel_ms(lambda: enc.encode_long(x64, None, Fx))
        if rank == 0:
            ab64 = algorithmic_bytes(False, False, False, coded_line_frac=coded)
            line["f64_layout"] = {"value": round(float(Fx) * HOP * world * args.steps / e64 / 1e6, 3), "unit": "Msamples/s",
                                  "frames": Fx, "ms_per_step": round(e64 / args.steps * 1e3, 4),
                                  "layout": "float64 signed fractions in, int32 mantissa plane out (round-1 layout)",
                                  "kernels": kernel_report(list(zip(KERNEL_NAMES, k64)), ab64, Fx, {})}
        del x64

    if not args.skip_extras:
        # ---- SURVEY.md 8(d) metric as defined: page-locked host PCM -> codes in page-locked host memory.  At N > 1 every rank
        # runs it at the same time on its own buffers (between barriers) and the slowest rank counts.
        h2h = host_to_host_leg(np, torch, enc, device, args.h2h_frames, first, barrier,
                               lambda v: max_over_ranks(v, coll_device), world)
        if rank == 0:
            line["host_to_host"] = h2h

    if world == 1 and not args.skip_extras:
        F = Fx                                             # (the configurations below run at the smaller batch)
        pcm = pcm[:(F + 1) * HOP].contiguous()
        # ---- configs[2]: stereo, joint M/S path, resident
        Fs = F // 2
        sl, sr = stream_slice(torch, device, "c3", 0, Fs)
        es = timed_steps(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        ks = kernel_ms(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        out = enc.encode_long(sl, sr, Fs, mantissa16=True)
        abj = algorithmic_bytes(True, True, True, coded_line_frac=coded_line_fraction(np, out, enc.h.bands(HOP, HOP)))
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

        # ---- configs[3]: block switching (long / start / 8 short / stop), mono, resident.  The block shapes are the
        # transient detector's on the content (pacfileThem.py:1025-1056, 1182-1214): its kernel runs inside the timed
        # step, the O(hops) sequencing of its peaks is host logic done once (the content does not change between steps)
        from mrcaudiocodec_amd import transient
        hops = F
        (xs,) = stream_slice(torch, device, "c4", 0, hops)
        sos = transient.design_sos(48000)
        peaks = torch.empty((hops, 1, HOP // 128 + 1), dtype=torch.float64, device=device)
        t0 = time.perf_counter()
        shp = transient.block_shape_array_dev(enc.h, xs.data_ptr(), 1, hops, 1, xs.numel(), sos)
        t_seq = time.perf_counter() - t0
        forced = c4_shapes(hops - 1)                                      # the cycle the content was built for
        groups, n_written = {}, 0
        for (a, b) in sorted({(int(r[1]), int(r[2])) for r in shp}):
            sel = shp[(shp[:, 1] == a) & (shp[:, 2] == b), 0]
            groups[(a, b)] = torch.from_numpy(np.ascontiguousarray(sel)).to(device)
        n_written = int(shp[:, 2].sum()) // HOP
        same_as_forced = all(k in forced and np.array_equal(np.asarray(forced[k]), groups[k].cpu().numpy()) for k in groups) \
            and len(forced) == len(groups)
        cur_stream = torch.cuda.current_stream(device).cuda_stream

        def run_detector():
            enc.h.dev_transient_peaks(hops, 1, sos, xs.data_ptr(), 1, xs.numel(), peaks.data_ptr(), cur_stream)

        def run_switched():
            run_detector()
            for (a, b), o in groups.items():
                enc.encode(a, b, xs, None, o.numel(), 0, o, mantissa16=True, offsets_checked=True)
        eb = timed_steps(run_switched)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            run_detector()
        ev1.record()
        torch.cuda.synchronize(device)
        det_ms = ev0.elapsed_time(ev1) / 3
        per_shape, rows_b = {}, []
        tb, tb_src = measured_traffic("switch")
        for (a, b), o in groups.items():
            km = kernel_ms(lambda: enc.encode(a, b, xs, None, o.numel(), 0, o, mantissa16=True, offsets_checked=True))
            bands_ab = enc.h.bands(a, b)
            nb = len(bands_ab)
            oo = enc.encode(a, b, xs, None, o.numel(), 0, o, mantissa16=True, offsets_checked=True)
            abk = algorithmic_bytes(False, True, True, a, b, nb, coded_line_frac=coded_line_fraction(np, oo, bands_ab))
            label = dict(KERNEL_LABEL)
            if (a, b) != (HOP, HOP):
                label["mdct"] = "mdct_wave_kernel"
                if (a, b) == (128, 128):
                    label["smr"] = "smr_short_kernel"
            rr = kernel_report(list(zip(KERNEL_NAMES, km)), abk, o.numel(), {}, label)   # (PMC traffic: per config, below)
            for r in rr:
                r["shape"] = "%dx%d" % (a, b)
            rows_b += rr
            per_shape["%dx%d" % (a, b)] = {"blocks": int(o.numel()), "kernel_ms": [round(float(v), 4) for v in km]}
        cfgs["block_switching"] = {
            "workload": "configs[3]: mono stream with a burst every 5th hop; block shapes from the transient detector on the "
                        "content -> (1024,1024), (1024,128), 7x(128,128), (128,1024) blocks, one launch set per shape; the "
                        "detector's kernel is part of the timed step",
            "value": round(float(n_written) * HOP * args.steps / eb / 1e6, 3), "unit": "Msamples/s", "hops": n_written,
            "ms_per_step": round(eb / args.steps * 1e3, 4), "per_shape": per_shape,
            "detector": {"kernel_ms": round(det_ms, 4), "peaks_to_shapes_host_s_once": round(t_seq, 3),
                         "shapes_equal_forced_cycle": bool(same_as_forced),
                         "what": "transient_peaks_kernel over %d hops (20th-order high-pass per hop from a zero state, one thread "
                                 "per hop) + the look-ahead sequencing of its peaks on the host" % hops},
            "roofline": roofline_of(rows_b, {"traffic_source": tb_src,
                                             "traffic": None if "smr_kernel" not in tb else round(tb["smr_kernel"] * hops),
                                             "traffic_note": "smr_kernel, all four block shapes of a step together"}),
            "hbm_traffic_bytes_per_hop": {k: v for k, v in tb.items()} or None, "kernels": rows_b}
        del peaks
        # ---- stream mode: MANY stereo streams, every stream's bit reservoir carried from block to block through the Huffman
        # savings (codecThem.py:224,274) -- the mode that writes the files the reference writes.  ONE call
        # (mrc_dev_encode_chained_pac): phase A batched over all blocks of all streams, the serial scan per stream on the
        # device, `.pac` files (headers, Close()'s block) packed in HBM
        cfgs["stream_mode"] = stream_mode_leg(np, torch, enc, device, 8192, 12, args.steps)
        # ---- ONE long stereo stream -> one `.pac` file (the reference's only real use case, pacfileThem.py:1064-1231)
        cfgs["single_stream"] = single_stream_leg(np, torch, enc, device, args.single_hops)
        line["configs"] = cfgs
        del xs, groups
        # ---- the per-GPU share of configs[4] at its real size, on this one GPU: rank 0's slice of the 8-rank job (frames
        # [0, 10^7 / 8) of the C3 stream, its halo in front), joint path -- the N = 1 anchor of the N = 8 line below
        del pcm
        enc._out.clear()
        torch.cuda.empty_cache()
        Fs = args.c4_frames
        sl, sr = stream_slices(torch, device, "c3", 0, Fs)
        torch.cuda.reset_peak_memory_stats(device)
        es = timed_steps(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        free_b, total_b = torch.cuda.mem_get_info(device)
        km = kernel_ms(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True), reps=2)
        line["configs4_share"] = {
            "workload": "configs[4], one rank's share at its real size: frames [0, %d) of the C3 stereo stream (10^7 / 8), joint "
                        "path, independent-frames mode, resident int16 PCM -> uint16 codes" % Fs,
            "value": round(2.0 * Fs * HOP * args.steps / es / 1e6, 3), "unit": "Msamples/s",
            "frames_per_step": Fs, "ms_per_step": round(es / args.steps * 1e3, 3),
            "kernel_ms": {n: round(float(v), 3) for n, v in zip(("mdct", "smr", "ms_switch", "bitalloc", "quantize"), km)},
            "hbm_in_use_GB": round((total_b - free_b) / 1e9, 2),
            "hbm_in_use_note": "device memory in use after the timed steps (hipMemGetInfo): the library's workspace -- MDCT "
                               "lines of four signals, SMRs, band peaks -- plus PCM and outputs held by the caller",
            "torch_peak_allocated_GB": round(torch.cuda.max_memory_allocated(device) / 1e9, 2)}
        del sl, sr
        enc._out.clear()
        torch.cuda.empty_cache()

    if world > 1 and not args.skip_extras:
        # ---- configs[4]: the C3 stereo stream, frame-sharded, joint path, host pack reported separately.  10^7 / 8 frames per
        # GPU (~54 GB of HBM: 10^7 frames at N = 8; the same per-GPU share at every N: weak scaling)
        del pcm
        enc._out.clear()
        torch.cuda.empty_cache()
        Fs = args.c4_frames
        first_s, cnt = shard_frames(world * Fs, world, rank)
        sl, sr = stream_slices(torch, device, "c3", first_s, Fs)
        es = timed_steps(lambda: enc.encode_long(sl, sr, Fs, mantissa16=True))
        out = enc.encode_long(sl, sr, Fs, mantissa16=True)
        pack = host_pack_rate(np, ppac, out, min(Fs, 16384), threads=max(1, min(16, (os.cpu_count() or 8) // world)))
        pack_rate = pack["huffman_priced_on_host_Msamples_s"]
        slowest = -max_over_ranks(-pack_rate, coll_device)            # min over ranks
        dpack = device_pack_rate(torch, enc, out, Fs, args.steps)
        dslow = -max_over_ranks(-dpack["huffman_priced_on_device_Msamples_s"], coll_device)
        del sl, sr, out
        torch.cuda.empty_cache()
        # ---- stream mode across ranks: whole streams per rank ("with many streams, shard by stream"), no
        # collective; the slowest rank's call counts
        sm = stream_mode_leg(np, torch, enc, device, 8192, 12, args.steps, first_stream=rank * 8192)
        sm_dt = max_over_ranks(sm["seconds_per_call"], coll_device)
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
            sm["workload"] = "x%d ranks, each: %s" % (world, sm["workload"])
            sm["value"] = round(2.0 * world * 8192 * 12 * HOP / sm_dt / 1e6, 3)
            sm["seconds_per_call"] = round(sm_dt, 5)
            sm["note"] += "; whole streams per rank, no collective, whole-job rate over the slowest rank's call"
            line["stream_mode"] = sm

    if rank == 0:
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
            line["speedup_vs_cpu_port"] = round(line["value"] / cpu_line["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def stream_mode_leg(np, torch, enc, device, nS, nT, steps, first_stream=0):
    """nS stereo streams of nT long blocks each (+ Close()), resident int16 PCM -> `.pac` files in HBM, one chained call."""
    gs = torch.Generator(device=device)
    gs.manual_seed(7 + first_stream)
    pl = torch.clamp(torch.round(torch.randn((nS, (nT + 1) * HOP), generator=gs, device=device, dtype=torch.float64) * 3000),
                     -32767, 32767)
    pl[:, :HOP] = 0
    pr = torch.clamp(torch.round(0.7 * pl + 0.3 * torch.roll(pl, 17, dims=1)), -32767, 32767)
    pr[:, :HOP] = 0
    ssl, ssr = pl.to(torch.int16).contiguous(), pr.to(torch.int16).contiguous()
    del pl, pr
    from mrcaudiocodec_amd import ChainSchedule
    one = np.array([(i * HOP, HOP, HOP) for i in range(nT)], dtype=np.int64)
    shapes_all = ChainSchedule([one] * nS)                  # the schedule in the C ABI's form, built once
    ns = np.full(nS, nT * HOP, dtype=np.uint32)
    r = enc.encode_chained_pac(ssl, ssr, shapes_all, num_samples=ns)                 # warm-up: buffers
    out_buf = torch.empty((int(r["total"]) + 4096,), dtype=torch.uint8, device=device)
    ts, ms = [], None
    for _ in range(max(3, steps)):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        r = enc.encode_chained_pac(ssl, ssr, shapes_all, num_samples=ns, out=out_buf)
        torch.cuda.synchronize(device)
        ts.append(time.perf_counter() - t0)
        if ms is None or ts[-1] == min(ts):
            ms = enc.h.chain_ms()
    dt = float(np.median(ts))
    res = {"workload": "%d stereo streams x %d chained joint long blocks + Close(), int16 PCM resident in HBM -> complete .pac "
                       "files in HBM; bit reservoirs carried from block to block on the device" % (nS, nT),
           "value": round(2.0 * nS * nT * HOP / dt / 1e6, 3), "unit": "Msamples/s", "seconds_per_call": round(dt, 5),
           "device_ms": {"phase_a_and_prep": round(float(ms[0]), 3), "serial_scan": round(float(ms[1]), 3),
                         "pack": round(float(ms[2]), 3)},
           "value_device_time_only": round(2.0 * nS * nT * HOP / (float(ms[3]) * 1e-3) / 1e6, 3),
           "pac_bytes_per_stereo_frame": round(r["total"] / (nS * nT), 1),
           "mean_final_reservoir_bits": round(float(np.mean(r["reservoir_out"])), 1),
           "note": "the call's wall time includes building and uploading the schedule of %d blocks on the host" % (nS * nT)}
    del ssl, ssr, out_buf
    return res


def single_stream_leg(np, torch, enc, device, hops):
    """ONE stereo stream of `hops` hops with bursts: detector -> block shapes -> one chained call.  Resident and host to
    host; the block-at-a-time loop (what this package did before the chained call, and what the reference does) timed on a
    prefix as the 'before'."""
    from mrcaudiocodec_amd import transient, pacfile as ppac, synth
    both = torch.empty((2, (hops + 1) * HOP), dtype=torch.int16, device=device)
    l, r = stream_slices(torch, device, "c5", 0, hops)
    both[0], both[1] = l, r
    del l, r
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    shp = transient.block_shape_array_dev(enc.h, both.data_ptr(), 1, hops, 2, both.shape[1])
    t_det = time.perf_counter() - t0
    last = len(shp)
    while last > 0 and shp[last - 1, 2] != HOP:                           # Close() wants a long last block
        last -= 1
    shp = shp[:last]
    samples = 2.0 * float(shp[:, 2].sum())
    n_short = int((shp[:, 1] + shp[:, 2] != 2 * HOP).sum())
    nsmp = [int(shp[:, 2].sum())]
    from mrcaudiocodec_amd import ChainSchedule
    sched = ChainSchedule([shp])
    rr = enc.encode_chained_pac(both[0:1], both[1:2], sched, num_samples=nsmp)       # warm-up
    out_buf = torch.empty((int(rr["total"]) + 4096,), dtype=torch.uint8, device=device)
    best, ms = None, None
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        rr = enc.encode_chained_pac(both[0:1], both[1:2], sched, num_samples=nsmp, out=out_buf)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, ms = dt, enc.h.chain_ms()
    host = both.cpu().numpy()
    enc.h.encode_chained_pac(host[0:1], host[1:2], [shp], num_samples=nsmp)
    t0 = time.perf_counter()
    hh = enc.h.encode_chained_pac(host[0:1], host[1:2], [shp], num_samples=nsmp)
    t_h2h = time.perf_counter() - t0
    same = hh["bytes"].tobytes() == rr["bytes"].cpu().numpy().tobytes()
    # before: the block-at-a-time loop on a prefix that ends with a long block
    k = min(64, len(shp))
    while k > 1 and shp[k - 1, 2] != HOP:
        k -= 1
    pre = [tuple(x) for x in shp[:k].tolist()]
    xf = synth.pcm_to_float(host[:, :int(pre[-1][0] + pre[-1][1] + pre[-1][2]) + HOP])
    ppac.encode_stereo_stream_per_block(enc.h, xf, pre[:1])
    t0 = time.perf_counter()
    ref = ppac.encode_stereo_stream_per_block(enc.h, xf, pre)
    t_loop = time.perf_counter() - t0
    pre_new = enc.h.encode_chained_pac(host[0:1], host[1:2], [pre], num_samples=[sum(b for (_, _, b) in pre)])
    loop_rate = 2.0 * sum(b for (_, _, b) in pre) / t_loop / 1e6
    res = {"workload": "ONE stereo 48 kHz stream of %d hops (noise floor + tone, a burst every 37th hop), block shapes from the "
                       "transient detector: %d blocks, %d of them short / transition; WAV samples -> one complete .pac file"
                       % (hops, len(shp), n_short),
           "value": round(samples / best / 1e6, 3), "unit": "Msamples/s", "what": "int16 PCM resident in HBM -> .pac bytes in HBM",
           "seconds_per_call": round(best, 5),
           "device_ms": {"phase_a_and_prep": round(float(ms[0]), 3), "serial_scan": round(float(ms[1]), 3),
                         "pack": round(float(ms[2]), 3)},
           "phase_b_us_per_block": round(1e3 * float(ms[1]) / (len(shp) + 2), 3),
           "host_to_host": {"value": round(samples / t_h2h / 1e6, 3), "unit": "Msamples/s", "bytes_equal_resident": bool(same),
                            "what": "int16 PCM in host memory -> .pac file in host memory (mrc_encode_chained_stream_pcm16_pac)"},
           "detector_seconds": round(t_det, 4),
           "before_per_block_loop": {"blocks": len(pre), "ms_per_block": round(1e3 * t_loop / len(pre), 4),
                                     "Msamples_s": round(loop_rate, 3),
                                     "bytes_equal_chained_call": bool(pre_new["bytes"].tobytes() == ref),
                                     "what": "pacfile.encode_stereo_stream_per_block: one mrc_encode_joint + host pack per block, "
                                             "the reservoir carried on the host (the round-2 path of cli.encode_wav)"},
           "speedup_vs_per_block_loop": round(samples / best / 1e6 / loop_rate, 1),
           "pac_bytes": int(rr["total"])}
    del both, out_buf
    return res


def host_to_host_leg(np, torch, enc, device, Fh, first_frame, barrier, max_ranks, world):
    """int16 PCM in page-locked host memory -> H2D -> kernels -> D2H -> codes in page-locked host memory
    (mrc_encode_stream_pcm16, 3 HIP streams), median of 5 runs per chunk size; then the same PCM to `.pac` chunk bytes
    (mrc_encode_stream_pcm16_pac).  Every rank measures between barriers; the times reported are the max over ranks."""
    from mrcaudiocodec_amd import PinnedArray
    keep = []

    def pin(shape, dt):
        p = PinnedArray(shape, dt); keep.append(p); return p.array
    host_pcm = pin(((Fh + 1) * HOP,), np.int16)
    piece = 1 << 17
    for f0 in range(0, Fh, piece):                          # generated on the device piece by piece (counter-based)
        nfr = min(piece, Fh - f0)
        (part,) = stream_slice(torch, device, "c2", first_frame + f0, nfr)
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
            barrier()
            t0 = time.perf_counter()
            enc.h.encode_stream_pcm16(host_pcm, None, None, chunk, outs)
            ts.append(max_ranks(time.perf_counter() - t0))
        runs[chunk] = float(np.median(ts))
    best = min(runs, key=runs.get)
    nchk = min(1 << 15, Fh)
    dev_pcm = torch.from_numpy(host_pcm[:(nchk + 1) * HOP].copy()).to(device)
    dev_out = enc.encode_long(dev_pcm, None, nchk, mantissa16=True)
    same = bool(np.array_equal(dev_out["mantissa"].cpu().numpy().view(np.uint16), outs["mantissa"][:nchk]))
    pcie = 2 * HOP + 2 * HOP + 2 * 4 * NB + 8
    res = {
        "value": round(world * Fh * HOP / runs[best] / 1e6, 3), "unit": "Msamples/s", "frames_per_gpu": Fh, "median_of": 5,
        "chunk_frames": best, "by_chunk_frames": {str(k): round(world * Fh * HOP / v / 1e6, 1) for k, v in runs.items()},
        "what": "int16 PCM in page-locked host memory -> H2D -> kernels -> D2H -> uint16 mantissas + int32 side info "
                "in page-locked host memory (mrc_encode_stream_pcm16, 3 HIP streams); every rank its own buffers, all ranks "
                "at once, whole-job rate over the slowest rank's time",
        "pcie_bytes_per_frame": pcie, "pcie_GBs_each_way_per_gpu": round(Fh * 2 * HOP / runs[best] / 1e9, 2),
        "pcie_ceiling": "page-locked copies on a one-GPU box (tools/pcie_rates.py): 55-57 GB/s one way alone, 25 / 47 GB/s "
                        "each way with both directions busy at 16 / 64 MiB per copy",
        "equals_resident_result": same}
    del dev_out, dev_pcm
    # ... and with the back end on the device as well: the same PCM -> `.pac` chunk bytes in page-locked host memory
    pac_buf = pin((Fh * HOP + 4096,), np.uint8)
    enc.h.encode_stream_pcm16_pac(host_pcm, None, None, True, best, {"bytes": pac_buf})
    ts = []
    for _ in range(5):
        barrier()
        t0 = time.perf_counter()
        pac = enc.h.encode_stream_pcm16_pac(host_pcm, None, None, True, best, {"bytes": pac_buf})
        ts.append(max_ranks(time.perf_counter() - t0))
    tp = float(np.median(ts))
    res["pac"] = {
        "value": round(world * Fh * HOP / tp / 1e6, 3), "unit": "Msamples/s", "chunk_frames": best,
        "bytes_per_frame": round(pac["bytes"].size / Fh, 1),
        "what": "the same PCM -> .pac chunk bytes of INDEPENDENT frames (no reservoir chaining: every frame starts from "
                "reservoir_in; Huffman pricing + bit packing on the device, mrc_encode_stream_pcm16_pac) in page-locked host "
                "memory.  The reference-equivalent WAV -> .pac rates are configs.stream_mode / configs.single_stream"}
    del pac
    for p in keep:
        p.free()
    return res


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
