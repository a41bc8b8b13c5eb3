#!/usr/bin/env python3
"""
Where the time of smr_kernel goes: shader-clock cycles per kernel phase, summed over all waves, from the
PROFILING build of the library (make -C mrcaudiocodec_amd/csrc OUT=.../libmrc_hip_prof.so BUILD=build_prof
EXTRA=-DMRC_PROFILE_PHASES).  Run as
    MRC_HIP_LIBRARY=mrcaudiocodec_amd/libmrc_hip_prof.so python tools/phase_profile.py [frames]
The timers serialise each phase (s_memtime + wait), so the total is a few % above the production kernel; the
FRACTIONS are what this is for.  Mono white noise as int16 PCM, long blocks (the bench workload).
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                                    # noqa: E402
from mrcaudiocodec_amd import _lib                              # noqa: E402
from mrcaudiocodec_amd.batch import StreamEncoder               # noqa: E402

NAMES = ["0 samples + Hann (wait for the loads)", "1 fft", "2 real split -> barrier", "3 compaction -> barrier", "4 masker table + searches",
         "5 scans done -> barrier", "6 chunk set-up", "7 far field (sorted sweep)", "8 node evaluation of a chunk",
         "9 the scans (rows | in-band | counts)", "10 ratio + band maxima", "11 wait for the other waves",
         "12 masker table -> barrier", "13 decision (slope range, node spacing)", "14 node terms", "15 node terms -> barrier",
         "16 prologue (arguments, unit, keys)", "17 peak flags + count", "18 counts -> barrier", "19 compaction + histogram zeroing",
         "20 real split + intensity", "21 masker: peak bin + its three intensities", "22 masker: level, frequency, Bark, intensity, constants",
         "23 masker: search hints (global)", "24 masker: the two walks + histogram"] + ["-"] * 7

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
enc = StreamEncoder(device_id=0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
x = torch.clamp(torch.round(torch.randn(((F + 1) * 1024,), generator=g, device=dev, dtype=torch.float64) * (0.1 * 32767)),
                -32767, 32767).to(torch.int16).contiguous()          # the bench's input: int16 PCM codes
fn = _lib.lib.mrc_debug_phase_cycles
fn.restype = C.c_int
buf = (C.c_ulonglong * 32)()
enc.encode_long(x, None, F)
assert fn(buf, 1) == 0
enc.encode_long(x, None, F)
assert fn(buf, 1) == 0
cyc = list(buf)[:len(NAMES)]
cyc[31] = 0
units = max(int(buf[31]), 1)                                    # workgroups sampled (every 64th)
tot = float(sum(cyc))
print(json.dumps({"frames": F, "workgroups_sampled": units, "wave_cycles_per_frame": round(tot / units, 1),
                  "phases": {n: {"cycles_per_frame": round(c / units, 1), "frac": round(c / tot, 4)} for n, c in zip(NAMES, cyc) if n != '-'}},
                 indent=1))
