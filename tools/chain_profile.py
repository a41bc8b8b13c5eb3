"""Phase cycles of chain_phase_b_kernel (profiling build: make -C mrcaudiocodec_amd/csrc OUT=.../libmrc_prof.so BUILD=build_prof
EXTRA=-DMRC_CHAIN_PROFILE; run with MRC_HIP_LIBRARY pointing at it).  One long stereo stream."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mrcaudiocodec_amd import Handle, synth, transient, _lib      # noqa: E402
from single_stream_bench import make_stream                        # noqa: E402

hops = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
period = int(sys.argv[2]) if len(sys.argv) > 2 else 37
h = Handle(device_id=0)
pcm = make_stream(hops, period)
shapes = transient.block_shapes(h, synth.pcm_to_float(pcm))
while shapes and shapes[-1][2] != 1024:
    shapes.pop()
raw = C.CDLL(_lib.LIB_PATH)
if len(sys.argv) > 3:
    h.set_option(4, int(sys.argv[3]))
prof = (C.c_ulonglong * 16)()
h.encode_chained_pac(pcm[0][None], pcm[1][None], [shapes], num_samples=[hops * 1024])
raw.mrc_debug_chain_profile(prof, 1)
h.encode_chained_pac(pcm[0][None], pcm[1][None], [shapes], num_samples=[hops * 1024])
raw.mrc_debug_chain_profile(prof, 1)
n = len(shapes) + 2
names = ["loop top", "regs -> LDS", "barrier 1", "issue next loads", "alloc head (cut, bits)", "alloc tail (batches)",
         "scale factors", "barrier 2", "quantise + price + wave sums", "barrier 3", "decision (thread 0)", "barrier 4"]
v = list(prof)
out = {"blocks": n, "cycles_per_block": {names[i]: v[i] / n for i in range(12)}, "total_cycles_per_block": sum(v[:12]) / n,
       "tail_events_per_block": v[12] / max(v[13], 1), "phase_b_ms": float(h.chain_ms()[1])}
print(json.dumps(out, indent=1))
