"""Exact mismatch counts of the HIP path against the oracle over a sweep (GPU box): the numbers behind
tests/parity.py::budget().  Writes one JSON object (profiles/r0N_parity_counts.json are copies of its output).

  python tools/parity_counts.py [n_fuzz] [out.json]

Sweep: n_fuzz seeded random small configurations (tests/test_gpu_parity.py::_fuzz_cases, a different seed than the
test), BASELINE config 2 (8 distinct 1080p bench frames), BASELINE config 3 (4096 x 4096, 6 octaves), the
adversarial content set at 640 x 480."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as oracle
from popsift_amd import capi
from popsift_amd.synth import synth, synth_float
from tests import adversarial as adv
from tests.parity import match_features, sort_iext
from tests.test_gpu_parity import _fuzz_cases, _wide_fuzz_cases

n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = sys.argv[2] if len(sys.argv) > 2 else None


def one(img, kw, acc):
    ref = oracle.run(oracle.default_config(**kw), img)
    ctx = capi.Context(capi.default_config(**kw))
    ctx.upload(img)
    ctx.extract()
    planes_ok = ctx.num_octaves == ref.num_octaves
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            planes_ok = planes_ok and np.array_equal(ctx.dump_plane(capi.PLANE_GAUSS, o, l), ref.gauss(o, l))
        a, b = sort_iext(ref.iext(o)), sort_iext(ctx.dump_iext(o))
        planes_ok = planes_ok and len(a) == len(b) and np.array_equal(a["xpos"], b["xpos"]) and np.array_equal(a["ypos"], b["ypos"])
    fb, db = ctx.download()
    fa, da = ref.features(), ref.descriptors()
    acc["cases"] += 1
    acc["plane_or_extrema_mismatch_cases"] += 0 if planes_ok and len(fa) == len(fb) else 1
    if len(fa) and len(fa) == len(fb):
        m = match_features(fa, da, fb, db, norm_scale=float(2 ** kw.get("norm_multi", 0)))
        acc["keypoints"] += len(fa); acc["descriptors"] += m["desc_compared"]
        for k in ("kp_miss", "ori_miss", "desc_miss"):
            acc[k] += m[k]
        acc["max_desc_dist"] = max(acc["max_desc_dist"], m["max_desc_dist"])
    ref.close(); ctx.close()


def fresh():
    return dict(cases=0, keypoints=0, descriptors=0, kp_miss=0, ori_miss=0, desc_miss=0, max_desc_dist=0.0,
                plane_or_extrema_mismatch_cases=0)


res = {}
acc = fresh()
for (w, h, s, is_float, kw) in _fuzz_cases(n_fuzz, 424242):
    one(synth_float(w, h, s) if is_float else synth(w, h, s), kw, acc)
res["fuzz_small_configs"] = acc
acc = fresh()
for (w, h, s, is_float, kw) in _wide_fuzz_cases(max(50, n_fuzz // 2), 171717):     # round 4: the whole Config space (modes, scale factors, grid filter)
    one(synth_float(w, h, s) if is_float else synth(w, h, s), kw, acc)
res["fuzz_wide_configs"] = acc
acc = fresh()
for i in range(8):
    one(synth(1920, 1080, 1000 + i), dict(octaves=5), acc)
res["config2_1080p_8_frames"] = acc
acc = fresh()
for i in range(8):
    one(synth(1920, 1080, 1000 + i), dict(octaves=5, sift_mode=2), acc)          # round 4: the mode the north_star quotes parity on
res["config2_1080p_8_frames_vlfeat_mode"] = acc
acc = fresh()
for i in range(4):                                                                 # round 4: the external caller's profile
    one((synth(1920, 1080, 1000 + i).astype(np.float32) / np.float32(256.0)).astype(np.float32),
        dict(octaves=5, filter_max_extrema=10000, grid_filter_mode=1, norm_multi=9), acc)
res["caller_profile_float_gridfilter_norm9_4_frames"] = acc
acc = fresh()
for (w, h, s_, fl, kw) in [(200, 150, 21, False, dict(octaves=4, upscale_factor=2.0, sift_mode=2)), (640, 480, 22, False, dict(octaves=3, upscale_factor=-2.0)),
                           (321, 243, 23, False, dict(octaves=4, upscale_factor=0.5, sift_mode=1)), (300, 200, 24, True, dict(octaves=4, upscale_factor=1.5, sift_mode=2)),
                           (513, 387, 25, False, dict(octaves=4, upscale_factor=-0.5))]:
    one(synth_float(w, h, s_) if fl else synth(w, h, s_), kw, acc)                 # round 4: scale factors other than -1 / 0 / +1
res["other_scale_factors"] = acc
acc = fresh()
one(np.ascontiguousarray(np.tile(synth(1920, 1080, 1000), (4, 3))[:4096, :4096]), dict(octaves=6), acc)
res["config3_4096x4096"] = acc
acc = fresh()
one(synth(4096, 4096, 3001), dict(octaves=6, sift_mode=2), acc)
res["config3_4096x4096_vlfeat_mode"] = acc
acc = fresh()
for name in sorted(adv.CONTENT):
    for kw in (dict(octaves=5), dict(octaves=5, sift_mode=2), dict(octaves=4, sift_mode=1, gauss_mode=3)):
        one(adv.make(name, 640, 480), kw, acc)
res["adversarial_640x480"] = acc
tot = fresh()
for v in res.values():
    for k in tot:
        tot[k] = max(tot[k], v[k]) if k == "max_desc_dist" else tot[k] + v[k]
res["total"] = tot
res["per_100k_keypoints"] = {k: round(tot[k] * 1e5 / max(1, tot["keypoints"]), 2) for k in ("kp_miss", "ori_miss", "desc_miss")}
txt = json.dumps(res, indent=1)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
