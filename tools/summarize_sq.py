#!/usr/bin/env python3
"""
SQ counter passes of tools/collect_counters.sh -> profiles/<tag>_sq_counters.json: per kernel the mean counter values
per launch, VALU instructions per frame and the fraction of the VALU issue capacity in use (a wave64 VALU
instruction occupies its SIMD for 4 cycles; 4 SIMDs x 256 CUs = 1024 SIMDs; GRBM_GUI_ACTIVE as reported by rocprofv3
on gfx950 is the SUM over the 8 XCDs, so the device runs GRBM_GUI_ACTIVE / 8 cycles).
Usage: summarize_sq.py <tag> <frames_per_launch> [gpurun_out]
"""
import collections, csv, glob, json, os, re, sys

tag, frames = sys.argv[1], int(sys.argv[2])
src = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
kern = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, tag + "_sq*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m or "mrc::" not in r["Kernel_Name"]:
                continue
            agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for (k, c), v in agg.items():
            kern[k][c] = sum(v) / len(v)
        for k, v in dur.items():
            kern[k].setdefault("dur_us_under_pmc", round(sum(v) / len(v), 1))
for k, c in kern.items():
    if "SQ_INSTS_VALU" in c:
        c["valu_insts_per_frame"] = round(c["SQ_INSTS_VALU"] / frames, 1)
        if "GRBM_GUI_ACTIVE" in c:
            c["valu_issue_frac_at_4cyc"] = round(c["SQ_INSTS_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_ACTIVE_INST_VALU counts quad-cycles with a VALU instruction executing, summed over the 1024 SIMDs
        c["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 3)
json.dump({"frames_per_launch": frames, "kernels": kern}, open(os.path.join(root, tag + "_sq_counters.json"), "w"), indent=1)
print(json.dumps(kern.get("smr_kernel", {}), indent=1))
