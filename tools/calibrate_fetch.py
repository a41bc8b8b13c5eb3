#!/usr/bin/env python3
"""
Calibration of rocprofv3's FETCH_SIZE for this library's access patterns (MI355X_MICROARCH.md, HBM section:
on gfx950 FETCH_SIZE can read exactly half of a wide coalesced stream; calibrate on a known byte count).
Runs the long-block MDCT kernel on NON-overlapped blocks (frame_stride = 2048: every sample is read by
exactly one frame, so the true read volume is n_frames * 16 KiB with no possible reuse) and on the
hop-overlapped stream (frame_stride = 1024).  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out> -- python3 tools/calibrate_fetch.py
and compare the counter of the two mdct_long_kernel dispatch groups with the known volumes printed here.
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mrcaudiocodec_amd import Handle

n = 65536
h = Handle()
dev = torch.device("cuda", 0)
x = torch.rand((n + 1) * 2048, dtype=torch.float64, device=dev) - 0.5
lines = torch.empty((n, 1024), dtype=torch.float64, device=dev)
scale = torch.empty((n,), dtype=torch.int32, device=dev)
for stride in (2048, 2048, 1024, 1024):
    h.dev_mdct(1024, 1024, n, x.data_ptr(), None, stride, None, lines.data_ptr(), scale.data_ptr())
    torch.cuda.synchronize()
print("blocks layout : true read bytes per dispatch =", n * 2048 * 8)
print("stream layout : algorithmic read bytes per dispatch =", (n + 1) * 1024 * 8)
