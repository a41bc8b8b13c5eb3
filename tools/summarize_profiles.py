#!/usr/bin/env python3
"""
Turn rocprofv3 output directories (gpurun_out/, scratch) into the small summaries committed under profiles/:
  <tag>_kernel_stats.csv   --kernel-trace --stats rows of this library's kernels
  <tag>_traffic.json       FETCH_SIZE / WRITE_SIZE per kernel (two separate --pmc passes), corrected as
                           MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2: wide coalesced reads are
                           tallied at half their bytes; calibrated here with tools/calibrate_fetch.py:
                           a known 1,073,741,824-byte read reports 571,000 KB), per launch and per frame.
Usage: summarize_profiles.py <tag> <frames_per_step> <stats_dir> <fetch_dir> <write_dir> [steps]
`steps` = encode steps the profiled command ran (warm-up included).  Given, the counters of ALL dispatches of a kernel
are summed and divided by it (a block-switched step launches each kernel once per block shape); omitted, the mean per
dispatch is taken (one launch per step).
"""
import collections, csv, glob, json, os, re, sys

tag, frames, stats_dir, fetch_dir, write_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 0
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")

def kname(s):
    m = re.search(r"(\w+_kernel)", s)
    return m.group(1) if m and "mrc::" in s else None

stats = glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0]
with open(stats) as f, open(os.path.join(root, tag + "_kernel_stats.csv"), "w") as o:
    for i, line in enumerate(f):
        if i == 0 or "mrc::" in line:
            o.write(line)

def counters(d):
    path = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        if k:
            agg[k].append(float(r["Counter_Value"]))
    return {k: (sum(v) / steps if steps else sum(v) / len(v)) for k, v in agg.items()}

fetch, write = counters(fetch_dir), counters(write_dir)
out = {"frames_per_launch": frames, "steps_profiled": steps or None, "unit_note": "FETCH_SIZE/WRITE_SIZE are KB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
       "kernels": {}}
for k in fetch:
    b = (2 * fetch[k] + write.get(k, 0.0)) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KB": round(fetch[k], 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1),
                         "hbm_bytes_per_launch": round(b), "hbm_bytes_per_frame": round(b / frames, 1)}
json.dump(out, open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
