"""Quick stage-time probe on one GPU (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
from popsift_amd.synth import synth

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
img = synth(w, h, 1000)
t = time.time(); ctx = capi.Context(capi.default_config(octaves=5)); print("create %.3fs" % (time.time() - t))
t = time.time(); ctx.upload(img); ctx.sync(); print("upload+resize %.3fs" % (time.time() - t))
ctx.enable_timers(True)
for it in range(5):
    t = time.time(); ctx.extract(); ctx.sync(); wall = time.time() - t
    st = ctx.stage_times()
    print("iter %d wall %.3f ms  stages: pyr %.3f ext %.3f ori %.3f desc %.3f  sum %.3f" % (it, wall * 1e3, *st, sum(st)))
print("counts", ctx.counts())
for o in range(2):
    for l in range(1, 6):
        ms, by = ctx.time_blur(o, l, 20)
        print("blur o%d l%d: %.4f ms  %.1f GB/s (8N)" % (o, l, ms, by / ms / 1e6))
ctx.enable_timers(False)
n = 50
t = time.time()
for i in range(n): ctx.extract()
ctx.sync(); dt = (time.time() - t) / n
print("single-stream %.3f ms/frame -> %.1f Mpix/s" % (dt * 1e3, w * h / dt / 1e6))
ctxs = [capi.Context(capi.default_config(octaves=5)) for _ in range(4)]
for c in ctxs: c.upload(img); c.extract(); c.sync()
t = time.time()
for i in range(n):
    for c in ctxs: c.extract()
for c in ctxs: c.sync()
dt = (time.time() - t) / (n * len(ctxs))
print("4-stream %.3f ms/frame -> %.1f Mpix/s" % (dt * 1e3, w * h / dt / 1e6))
for nc in (2, 8, 16):
    ctxs = [capi.Context(capi.default_config(octaves=5)) for _ in range(nc)]
    for c in ctxs: c.upload(img); c.extract(); c.sync()
    t = time.time()
    for i in range(n):
        for c in ctxs: c.extract()
    for c in ctxs: c.sync()
    dt = (time.time() - t) / (n * len(ctxs))
    print("%d-stream %.3f ms/frame -> %.1f Mpix/s" % (nc, dt * 1e3, w * h / dt / 1e6))
    for c in ctxs: c.close()
