#!/usr/bin/env python3
"""
Where the time of smr_kernel goes: shader-clock cycles per kernel phase, summed over all waves, from the
PROFILING build of the library (make -C mrcaudiocodec_amd/csrc OUT=.../libmrc_hip_prof.so BUILD=build_prof
EXTRA=-DMRC_PROFILE_PHASES).  Run as
    MRC_HIP_LIBRARY=mrcaudiocodec_amd/libmrc_hip_prof.so python tools/phase_profile.py [frames]
The timers serialise each phase (s_memtime + wait), so the total is a few % above the production kernel; the
FRACTIONS are what this is for.  Mono white noise, long blocks (the bench workload).
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                                    # noqa: E402
from mrcaudiocodec_amd import _lib                              # noqa: E402
from mrcaudiocodec_amd.batch import StreamEncoder               # noqa: E402

NAMES = ["load+hann", "fft", "real split + intensity", "peak scan + compaction", "masker table + searches",
         "suffix/prefix sums + count scan", "chunk setup", "far field", "direct pairs", "partial pairs",
         "in-band + lower + log10 + band max", "final store"]

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
enc = StreamEncoder(device_id=0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
p = torch.clamp(torch.round(torch.randn(((F + 1) * 1024,), generator=g, device=dev, dtype=torch.float64) * (0.1 * 32767)),
                -32767, 32767)
x = (torch.sign(p) * 2.0 * torch.abs(p) / 65535).contiguous()
fn = _lib.lib.mrc_debug_phase_cycles
fn.restype = C.c_int
buf = (C.c_ulonglong * 16)()
enc.encode_long(x, None, F)
assert fn(buf, 1) == 0
enc.encode_long(x, None, F)
assert fn(buf, 1) == 0
cyc = list(buf)[:len(NAMES)]
tot = float(sum(cyc))
print(json.dumps({"frames": F, "wave_cycles_per_frame": round(tot / F, 1),
                  "phases": {n: {"cycles_per_frame": round(c / F, 1), "frac": round(c / tot, 4)} for n, c in zip(NAMES, cyc)}},
                 indent=1))
