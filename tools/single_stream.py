"""One context, N sequential frames: the workload used for rocprofv3 kernel-trace profiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5))
ctx.upload(img)
for i in range(n):
    ctx.extract()
    ctx.sync()
print(ctx.counts())
