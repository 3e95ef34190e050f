"""dev tool (GPU box): stage times of one 1080p bench frame per DescMode (event timers, one context, median of 9):
pyramid / extrema / orientation / descriptors in ms.   python tools/desc_modes_ms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
img = synth(1920, 1080, 1000)
for name, mode in (("loop", 0), ("iloop", 1), ("grid", 2), ("igrid", 3), ("notile", 4)):
    ctx = capi.Context(capi.default_config(octaves=5, sift_mode=2, desc_mode=mode))
    ctx.upload(img)
    for _ in range(3):
        ctx.extract(); ctx.counts()
    ctx.enable_timers(True)
    st = []
    for _ in range(9):
        ctx.extract(); st.append(ctx.stage_times())
    med = [sorted(s[i] for s in st)[4] for i in range(4)]
    print("%-7s descriptors %.4f ms   (pyramid %.3f extrema %.3f orientation %.3f)  %s" % (name, med[3], med[0], med[1], med[2], ctx.counts()))
    ctx.close()
