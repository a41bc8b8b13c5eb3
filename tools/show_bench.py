"""Prints the figures of a bench.py JSON line one per row (reading aid for gpurun logs)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "n_gpus", d["n_gpus"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"],
      "coded", d.get("coded_line_fraction"))
for k in d["kernels"]:
    print("   ", k["name"], k["ms"], "ms", k["frac_hbm"])
if "roofline" in d and d["roofline"].get("valu"):
    print("   valu", d["roofline"]["valu"])
if "f64_layout" in d:
    print("f64_layout", d["f64_layout"]["value"])
if "host_to_host" in d:
    print("h2h", d["host_to_host"]["value"], d["host_to_host"]["by_chunk_frames"], "pac", d["host_to_host"]["pac"]["value"])
c = d.get("configs", {})
if "stereo_ms" in c:
    print("stereo", c["stereo_ms"]["value"], c["stereo_ms"]["ms_per_step"])
    for k in c["stereo_ms"]["kernels"]:
        print("   ", k["name"], k["ms"], "ms", k["frac_hbm"])
    print("    pack", c["stereo_ms"].get("host_pack"), c["stereo_ms"].get("device_pack"))
if "block_switching" in c:
    b = c["block_switching"]
    print("switch", b["value"], b["ms_per_step"], b.get("detector"))
    for k in b["kernels"]:
        print("   ", k.get("shape"), k["name"], k["ms"], "ms", k["frac_hbm"])
for name in ("stream_mode", "single_stream"):
    if name in c:
        print(name, json.dumps(c[name]))
for name in ("configs4", "stream_mode"):
    if name in d:
        print(name, json.dumps(d[name]))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("vectorised_port_value"), d["cpu_baseline"]["all_cores"]["value"])
