import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img); ctx.extract(); ctx.sync()
tot = 0
for l in range(1, 6):
    ctx.time_blur(0, l, 10)
    ms, by = ctx.time_blur(0, l, 100); tot += ms
    print("o0 l%d %.2f us %.0f GB/s" % (l, ms * 1e3, by / ms / 1e6), end=" | ")
print("sum %.1f us  avg %.0f GB/s" % (tot * 1e3, 5 * by / tot / 1e6))
