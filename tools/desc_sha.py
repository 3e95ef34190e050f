"""dev tool (GPU box): SHA-1 of the descriptors and features of a few frames -- to check that a restructured kernel is
bit-identical to the previous build (the histogram adds are integer, so a change of the visiting order must not move a bit).
usage: python tools/desc_sha.py"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
from popsift_amd.synth import synth
h = hashlib.sha1()
n = 0
out = []
for (w, hh, seed, kw) in [(1920, 1080, 1000, dict(octaves=5, sift_mode=2)), (1920, 1080, 7, dict(octaves=5)),
                          (640, 480, 3, dict(octaves=4, sift_mode=1, norm_mode=1, norm_multi=9)), (333, 251, 5, dict(upscale_factor=0.0)),
                          (800, 600, 9, dict(octaves=4, sigma=2.0, levels=2))]:
    ctx = capi.Context(capi.default_config(**kw)); ctx.upload(synth(w, hh, seed)); ctx.extract()
    f, d = ctx.download()
    order = np.lexsort((f["ypos"], f["xpos"], f["sigma"]))
    for k in ("xpos", "ypos", "sigma", "num_ori", "orientation"):           # desc_idx depends on the candidate order of the run
        h.update(np.ascontiguousarray(f[k][order]).tobytes())
    # descriptors in feature order
    for i in order:
        for k in range(int(f["num_ori"][i])):
            h.update(d[int(f["desc_idx"][i][k])].tobytes())
            out.append(d[int(f["desc_idx"][i][k])])
    n += len(d); ctx.close()
if len(sys.argv) > 1:
    np.save(sys.argv[1], np.array(out))
print("descriptors", n, "sha1", h.hexdigest())
