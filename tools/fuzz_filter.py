"""dev tool (GPU box): random grid-filter configurations, HIP vs oracle (scale-ordered modes: exact survivor sets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as oracle
from popsift_amd import capi
from popsift_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for i in range(n):
    w, h = int(rng.integers(200, 700)), int(rng.integers(150, 500))
    img = synth(w, h, 5000 + i)
    base = oracle.run(oracle.default_config(octaves=4), img)
    if base.ext_total < 50: continue
    kw = dict(octaves=4, filter_max_extrema=int(base.ext_total * rng.uniform(0.1, 0.85)),
              filter_grid_size=int(rng.integers(1, 9)), grid_filter_mode=int(rng.integers(1, 3)))
    ref = oracle.run(oracle.default_config(**kw), img)
    ctx = capi.Context(capi.default_config(**kw)); ctx.upload(img); ctx.extract()
    f, d = ctx.download()
    ok = len(f) == ref.ext_total
    for o in range(ref.num_octaves):
        a = ref.iext(o); b = ctx.dump_iext(o)
        ka = sorted((float(e["xpos"]), float(e["ypos"]), int(e["lpos"]), int(e["ignore"])) for e in a)
        kb = sorted((float(e["xpos"]), float(e["ypos"]), int(e["lpos"]), int(e["ignore"])) for e in b)
        ok = ok and ka == kb
    if not ok:
        bad += 1
        print("MISMATCH", w, h, kw, len(f), ref.ext_total, base.ext_total)
    ctx.close()
print("grid-filter cases", n, "mismatches", bad)
