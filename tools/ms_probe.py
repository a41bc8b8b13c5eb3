"""Experiment aid: kernel times of one joint encode_long call under a library build given by MRC_HIP_LIBRARY."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd.batch import StreamEncoder
enc = StreamEncoder(device_id=0)
F = 65536
g = torch.Generator(device=enc.device); g.manual_seed(1)
pl = torch.clamp(torch.round(torch.randn(((F + 1) * 1024,), generator=g, device=enc.device, dtype=torch.float64) * 3000), -32767, 32767).to(torch.int16)
pr = torch.roll(pl, 333)
enc.h.set_timing(True)
for _ in range(3):
    enc.encode_long(pl, pr, F, mantissa16=True)
    print([round(float(x), 4) for x in enc.h.kernel_ms()])
