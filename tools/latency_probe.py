"""dev tool: wall time of one frame through psx_extract without stage timers (single context), and the
bench-style throughput with N contexts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img)
for _ in range(5): ctx.extract(); ctx.sync()
ts = []
for _ in range(30):
    t = time.perf_counter(); ctx.extract(); ctx.sync(); ts.append(time.perf_counter() - t)
ts.sort()
print("single context: median %.3f ms  min %.3f ms" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
