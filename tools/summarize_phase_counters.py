#!/usr/bin/env python3
"""gpurun_out/<tag>_<name>_pc (tools/phase_counters.sh) -> table of smr_kernel counters per build, per frame.
Usage: summarize_phase_counters.py <tag> <frames> [out.txt]"""
import collections, csv, glob, os, re, sys
tag, frames = sys.argv[1], int(sys.argv[2])
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
rows = {}
for d in sorted(glob.glob(os.path.join(src, tag + "_*_pc"))):
    name = os.path.basename(d)[len(tag) + 1:-3]
    agg, dur = collections.defaultdict(list), []
    for path in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if "smr_kernel" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if agg:
        rows[name] = {k: sum(v) / len(v) / frames for k, v in agg.items()}
        rows[name]["us"] = sum(dur) / len(dur)
cols = ["us", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
        "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES"]
lines = ["per frame (%d frames per launch; us = kernel duration under the counters)" % frames,
         "%-10s" % "build" + "".join("%22s" % c for c in cols)]
order = ["stop0", "stop1", "stop2", "stop3", "stop4", "stop5", "nosweep", "half1", "default"]
for name in sorted(rows, key=lambda n: order.index(n) if n in order else 99):
    lines.append("%-10s" % name + "".join("%22.1f" % rows[name].get(c, float("nan")) for c in cols))
text = "\n".join(lines)
print(text)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
