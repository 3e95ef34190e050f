"""dev tool: u8 vs float32 input frames (AliceVision passes float images)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
from popsift_amd.synth import synth, synth_float
for name, img in (("u8", synth(1920, 1080, 1000)), ("f32", synth_float(1920, 1080, 1000))):
    ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img)
    for _ in range(3): ctx.extract(); ctx.sync()
    ctx.enable_timers(True)
    ts = []
    for _ in range(10):
        t = time.perf_counter(); ctx.extract(); ctx.sync(); ts.append(time.perf_counter() - t)
    print(name, "wall %.3f ms" % (sorted(ts)[5] * 1e3), "stages", ["%.3f" % v for v in ctx.stage_times()], ctx.counts())
    ctx.close()
