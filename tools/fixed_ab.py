"""dev tool (GPU box): the fixed-span Gauss modes (Fixed9 / Fixed15) with the one-kernel-per-octave path (pyramid_fixed.hip)
against the per-level kernels (POPSIFT_FIXED_FUSED=0): SHA-1 over all Gaussian planes, keypoints, stage medians (event timers).
usage: python tools/fixed_ab.py [w h [frames]]     (run once per setting of POPSIFT_FIXED_FUSED; compare the lines)"""
import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
from popsift_amd.synth import synth, synth_float
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for name, kw in [("default", {}), ("fixed9", dict(gauss_mode=4)), ("fixed15", dict(gauss_mode=5)), ("fixed9_float", dict(gauss_mode=4)),
                 ("relative", dict(gauss_mode=1)), ("relative_float", dict(gauss_mode=1)), ("relative_s2l4", dict(gauss_mode=1, levels=4, sigma=2.0))]:
    img = synth_float(w, h, 3) if name.endswith("float") else synth(w, h, 3)
    ctx = capi.Context(capi.default_config(octaves=5, **kw)); ctx.upload(img)
    ctx.enable_timers(True)
    rows = []
    for i in range(n + 3):
        ctx.extract(); ctx.sync()
        if i >= 3: rows.append(ctx.stage_times())
    sha = hashlib.sha1()
    for o in range(ctx.num_octaves):
        for l in range(ctx.num_levels):
            sha.update(np.ascontiguousarray(ctx.dump_plane(capi.PLANE_GAUSS, o, l)).tobytes())
    t = time.perf_counter()
    for i in range(10): ctx.extract(); nk = len(ctx.download()[0])
    wall = (time.perf_counter() - t) * 100
    print("%-13s fused=%s planes %s  %6d keypoints  frame %.3f ms  stages(ms) %s" % (
        name, os.environ.get("POPSIFT_FIXED_FUSED", "1") + os.environ.get("POPSIFT_INTERP_FUSED", "1"), sha.hexdigest()[:16], nk, wall,
        [round(float(v), 4) for v in np.median(np.array(rows), axis=0)]))
    ctx.close()
