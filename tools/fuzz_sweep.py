"""dev tool (GPU box): wider version of tests/test_gpu_parity.py::test_fuzz_small_configs.
usage: python tools/fuzz_sweep.py [n] [seed] [size-scale] [wide]
"wide" walks the whole Config space (every GaussMode / ScalingMode, fractional scale factors, grid filter:
tests/test_gpu_parity.py::_wide_fuzz_cases) instead of the default-branch sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as oracle
from popsift_amd import capi
from popsift_amd.synth import synth, synth_float
from tests.parity import match_features, sort_iext
from tests.test_gpu_parity import _fuzz_cases, _wide_fuzz_cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 4 and sys.argv[4] == "wide":
    _fuzz_cases = _wide_fuzz_cases
bad = 0
worst = 1.0
nkp = 0
for (w, h, s, is_float, kw) in _fuzz_cases(n, seed):
    w, h = w * scale, h * scale
    img = synth_float(w, h, s) if is_float else synth(w, h, s)
    ref = oracle.run(oracle.default_config(**kw), img)
    ctx = capi.Context(capi.default_config(**kw)); ctx.upload(img); ctx.extract()
    ok = ctx.num_octaves == ref.num_octaves and ctx.num_levels == ref.num_levels
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            ok = ok and np.array_equal(ctx.dump_plane(capi.PLANE_GAUSS, o, l), ref.gauss(o, l))
        a, b = sort_iext(ref.iext(o)), sort_iext(ctx.dump_iext(o))
        ok = ok and len(a) == len(b) and np.array_equal(a["xpos"], b["xpos"]) and np.array_equal(a["ypos"], b["ypos"])
    fb, db = ctx.download()
    fa, da = ref.features(), ref.descriptors()
    ok = ok and len(fa) == len(fb)
    nkp += len(fa)
    if ok and len(fa):
        m = match_features(fa, da, fb, db, norm_scale=float(2 ** kw["norm_multi"]))
        worst = min(worst, m["ori_match"], m["desc_match"])
        if m["kp_match"] < 1.0 or m["ori_match"] < 1.0 or m["desc_match"] < 1.0:
            print("imperfect", w, h, s, is_float, kw, {k: m[k] for k in ("n_a", "kp_match", "ori_match", "desc_match", "max_desc_dist")})
    if not ok:
        bad += 1
        print("MISMATCH", w, h, s, is_float, kw)
    ctx.close()
print("cases", n, "keypoints", nkp, "hard mismatches", bad, "worst ori/desc match fraction", worst)
