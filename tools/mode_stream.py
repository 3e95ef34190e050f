"""One context, N sequential 1080p frames in a given configuration: the workload for rocprofv3 kernel traces of the
non-default modes.  usage: python tools/mode_stream.py [n] key=value ...   (e.g. gauss_mode=4 desc_mode=3)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v else int(v)
w, h = int(kw.pop("w", 1920)), int(kw.pop("h", 1080))
ctx = capi.Context(capi.default_config(octaves=int(kw.pop("octaves", 5)), **kw))
ctx.upload(synth(w, h, 1000))
for i in range(n):
    ctx.extract()
    ctx.sync()
print(kw, ctx.counts())
