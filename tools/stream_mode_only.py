import sys, json, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from mrcaudiocodec_amd.batch import StreamEncoder
enc = StreamEncoder(device_id=0)
if len(sys.argv) > 1:
    enc.h.set_option(4, int(sys.argv[1]))
print(json.dumps(bench.stream_mode_leg(np, torch, enc, enc.device, 8192, 12, 5)))
