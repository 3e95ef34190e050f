"""dev tool: median stage times (ms) of one context on the 1080p bench frame, 40 extractions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img)
ctx.enable_timers(True)
for _ in range(5): ctx.extract()
ts = []
for _ in range(40):
    ctx.extract(); ts.append(ctx.stage_times())
med = [sorted(t[i] for t in ts)[len(ts) // 2] for i in range(len(ts[0]))]
print(" ".join("%.4f" % m for m in med))
