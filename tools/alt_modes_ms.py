"""dev tool (GPU box): wall time per 1080p frame (extract + download, one context) of the non-default pyramid branches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(1920, 1080, 3)
for name, kw in [("default", {}), ("relative_all", dict(gauss_mode=2)), ("scale_direct", dict(scaling_mode=0)), ("relative", dict(gauss_mode=1)),
                 ("fixed9", dict(gauss_mode=4)), ("fixed15", dict(gauss_mode=5)), ("up0.5", dict(upscale_factor=0.5)), ("up1.5", dict(upscale_factor=1.5)),
                 ("down0.5", dict(upscale_factor=-0.5))]:
    ctx = capi.Context(capi.default_config(octaves=5, **kw)); ctx.upload(img)
    for i in range(3): ctx.extract(); ctx.download()
    t = time.perf_counter()
    for i in range(10): ctx.extract(); n = len(ctx.download()[0])
    print("%-14s %.3f ms  %d keypoints" % (name, (time.perf_counter() - t) * 100, n)); ctx.close()
