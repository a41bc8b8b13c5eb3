"""filter of `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel (see kernel_resources.sh)"""
import re
import subprocess
import sys

cur = {}
for l in sys.stdin:
    if "error" in l or "warning:" in l:
        print(l.rstrip())
    m = re.search(r"remark: +([A-Za-z \[\]/]+): (.*?) \[-Rpass", l)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS Size"):
        d = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d).replace("mrc::(anonymous namespace)::", "").replace("void ", "")
        print("%-52s VGPR %3s AGPR %3s spill %3s scratch %5s occ %s LDS %6s" % (
            d[:52], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"),
            cur.get("Occupancy [waves/SIMD]"), v))
