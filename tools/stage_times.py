"""dev tool (GPU box): median stage times (pyramid, extrema, orientation, descriptors; event timers) of one context over N
sequential 1080p frames.  usage: python tools/stage_times.py [frames] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
from popsift_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = capi.Context(capi.default_config(octaves=5, sift_mode=2))
ctx.upload(synth(1920, 1080, seed))
ctx.enable_timers(True)
rows = []
for i in range(n + 5):
    ctx.extract(); ctx.sync()
    if i >= 5:
        rows.append(ctx.stage_times())
print("counts", ctx.counts(), "stage medians (ms)", [round(float(v), 4) for v in np.median(np.array(rows), axis=0)])
