#!/usr/bin/env python3
"""Page-locked host <-> device copy rates of the box (the ceiling of bench.py's host_to_host figure): H2D alone, D2H alone,
both directions at once on two streams, for a few transfer sizes.  Usage: python tools/pcie_rates.py"""
import time, torch
dev = torch.device("cuda", 0)
for mb in (16, 64, 256):
    n = mb << 20
    h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_in = torch.empty(n, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run(h2d, d2h, reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_in.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_out.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
        return n * reps / (time.perf_counter() - t0) / 1e9
    run(True, True, 3)
    print("%4d MiB: H2D alone %.1f GB/s, D2H alone %.1f GB/s, both at once %.1f GB/s each way"
          % (mb, run(True, False), run(False, True), run(True, True)), flush=True)
