"""Device time of transient_peaks_kernel over a mono int16 stream of `hops` hops (torch events on the launch stream)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrcaudiocodec_amd import transient             # noqa: E402
from mrcaudiocodec_amd.batch import StreamEncoder   # noqa: E402

hops = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
enc = StreamEncoder(device_id=0)
dev = enc.device
x = (torch.randn(((hops + 1) * 1024,), device=dev) * 300).to(torch.int16)
sos = transient.design_sos(48000)
peaks = torch.empty((hops, 1, 9), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
enc.h.dev_transient_peaks(hops, 1, sos, x.data_ptr(), 1, x.numel(), peaks.data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    enc.h.dev_transient_peaks(hops, 1, sos, x.data_ptr(), 1, x.numel(), peaks.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
print("transient_peaks_kernel: %.4f ms per %d hops, checksum %.6f" % (e0.elapsed_time(e1) / 5, hops, float(peaks.sum().item())))
