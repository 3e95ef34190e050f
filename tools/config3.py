"""BASELINE config 3: 4096x4096 frame, 6 octaves, upscale x2 (8192x8192 octave 0): HBM-bound pyramid stress."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(4096, 4096, 1000)
ctx = capi.Context(capi.default_config(octaves=6))
ctx.upload(img); ctx.extract(); ctx.sync()
ctx.enable_timers(True)
for it in range(3):
    t = time.time(); ctx.extract(); ctx.sync(); wall = time.time() - t
    st = ctx.stage_times()
    print("wall %.2f ms stages pyr %.2f ext %.2f ori %.2f desc %.2f -> %.0f Mpix/s" % (wall * 1e3, *st, 4096 * 4096 / wall / 1e6))
print("counts", ctx.counts(), "octave0", ctx.octave_dims(0))
f, d = ctx.download()
nrm = np.sqrt((d.astype(np.float64) ** 2).sum(1))
print("desc norm ok", bool(np.all(np.abs(nrm - 1) < 1e-4)), "num_ori sum == ndesc", int(f["num_ori"].sum()) == len(d))
for l in range(1, 6):
    ms, by = ctx.time_blur(0, l, 10)
    print("blur o0 l%d %.1f us %.0f GB/s" % (l, ms * 1e3, by / ms / 1e6), end=" | ")
print()
