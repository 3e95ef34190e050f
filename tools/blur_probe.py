"""dev tool: k_blur launch durations, octave 0 of a 1080p frame (levels 1..5 = radii 5,7,8,10,13)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img); ctx.extract(); ctx.sync()
for rep in range(3):
    tot = 0
    out = []
    for l in range(1, 6):
        ctx.time_blur(0, l, 10)
        ms, by = ctx.time_blur(0, l, 100); tot += ms
        out.append("l%d %.2f us" % (l, ms * 1e3))
    print(" | ".join(out), "| sum %.1f us  avg %.0f GB/s" % (tot * 1e3, 5 * by / tot / 1e6))
