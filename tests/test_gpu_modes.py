"""GPU parity of the alternative pyramid and descriptor modes: Config::setGaussMode
(VLFeat_Relative, VLFeat_Relative_All, Fixed9, Fixed15), setScalingMode(ScaleDirect), setDescMode(iloop, grid, igrid,
notile).  Each mode is a different numerical result with its own oracle branch (pinned against the reference's own
kernels in tests/test_ref_shim_cpu.py) and its own reference-generated fixture (tests/golden/ref_mode_*.npz)."""
import numpy as np
import pytest

from popsift_amd.synth import synth, synth_float
from tests import golden_util as gu
from tests.parity import assert_descriptor_rows, extrema_from_features, assert_parity, budget, match_features, sort_iext

pytestmark = pytest.mark.gpu

ALT_PYRAMIDS = [
    dict(gauss_mode=1), dict(gauss_mode=2), dict(gauss_mode=4), dict(gauss_mode=5),
    dict(scaling_mode=0), dict(scaling_mode=0, gauss_mode=1), dict(scaling_mode=0, gauss_mode=4),
    dict(scaling_mode=0, gauss_mode=3, sift_mode=1), dict(scaling_mode=0, gauss_mode=2),
    dict(gauss_mode=1, upscale_factor=0.0, sift_mode=1), dict(gauss_mode=2, upscale_factor=-1.0, sift_mode=2),
    dict(gauss_mode=1, levels=4, sigma=1.4), dict(gauss_mode=5, upscale_factor=0.0),
]


@pytest.mark.parametrize("kw", ALT_PYRAMIDS)
def test_alternative_pyramid_modes(oracle, capi, kw):
    """Planes bit-exact, initial extrema identical, features within the mismatch budget, for every non-default branch
    of build_pyramid; byte and float input."""
    for is_float, (w, h) in ((False, (333, 251)), (True, (160, 120))):
        img = synth_float(w, h, 21) if is_float else synth(w, h, 21)
        cfg = dict(octaves=4, **kw)
        ref = oracle.run(oracle.default_config(**cfg), img)
        ctx = capi.Context(capi.default_config(**cfg))
        ctx.upload(img)
        ctx.extract()
        assert ctx.num_octaves == ref.num_octaves and ctx.num_levels == ref.num_levels
        for o in range(ref.num_octaves):
            for l in range(ref.num_levels):
                g = ctx.dump_plane(capi.PLANE_GAUSS, o, l)
                assert np.array_equal(g.view(np.uint32), ref.gauss(o, l).view(np.uint32)), \
                    "%s plane (%d,%d): max abs err %g" % (kw, o, l, np.abs(g - ref.gauss(o, l)).max())
            a, b = sort_iext(ref.iext(o)), sort_iext(ctx.dump_iext(o))
            assert len(a) == len(b)
            for f in ("xpos", "ypos", "lpos"):
                assert np.array_equal(a[f], b[f]), (kw, o, f)
        fb, db = ctx.download()
        assert len(fb) == ref.ext_total
        if len(fb):
            assert_parity(match_features(ref.features(), ref.descriptors(), fb, db), what=str(kw), **budget(len(fb)))
        ctx.close()


@pytest.mark.parametrize("desc_mode,name", [(1, "iloop"), (3, "igrid"), (4, "notile")])
def test_interpolating_descriptor_modes(oracle, capi, desc_mode, name):
    """iloop / igrid / notile: bilinear gradients in the keypoint frame; descriptors within 1e-3 of the oracle's
    (both normalisations), and different from "loop"."""
    img = synth(480, 360, 8)
    for norm in (dict(), dict(norm_mode=1, norm_multi=9)):
        cfg = dict(octaves=4, desc_mode=desc_mode, **norm)
        ref = oracle.run(oracle.default_config(**cfg), img)
        ctx = capi.Context(capi.default_config(**cfg))
        ctx.upload(img)
        ctx.extract()
        fb, db = ctx.download()
        assert len(fb) == ref.ext_total > 500
        m = match_features(ref.features(), ref.descriptors(), fb, db, norm_scale=float(2 ** norm.get("norm_multi", 0)))
        print(name, {k: v for k, v in m.items() if k != "misses"})
        assert_parity(m, what=name, **budget(len(fb)))
        ctx.close()
    loop = oracle.run(oracle.default_config(octaves=4, norm_mode=1, norm_multi=9), img)
    assert np.abs(loop.descriptors() - ref.descriptors()).max() > 1.0


def test_grid_descriptor_mode(oracle, capi):
    """grid: sample points are snapped to pixels through (int)(pt + (round(pt + pix) - pt)) (s_desc_grid.cu:72-78); for
    |round| < |pt|/2 the sum can land one ulp below the integer and truncate to the neighbouring pixel, decided by the
    last bits of pt -- of the keypoint position and of sin / cos of the orientation.  Both sides evaluate sin / cos in
    double and round once (round 3), so EQUAL orientation bits give equal sample positions: the descriptor stage is
    checked strictly, on the device's own oriented extrema (the oracle redoes the descriptors for exactly those),
    with the ordinary budget.  End to end the orientations themselves differ in their last bits (transcendentals in
    the orientation histogram), which moves some knife-edge samples: that comparison is bounded, not strict."""
    img = synth(480, 360, 8)
    cfg = dict(octaves=4, desc_mode=2)
    ref = oracle.run(oracle.default_config(**cfg), img)
    ctx = capi.Context(capi.default_config(**cfg))
    ctx.upload(img)
    ctx.extract()
    fb, db = ctx.download()
    ext = ctx.dump_extrema()
    assert len(fb) == ref.ext_total > 500 and len(ext) == len(fb)
    # strict: same keypoints, same orientation bits -> same descriptors within 1e-3
    want = ref.describe(ext, len(db))
    worst = assert_descriptor_rows(want, db, len(fb), what="grid descriptor stage")
    print("grid stage-level max L2 %.3g over %d descriptors" % (worst, len(db)))
    # end to end (bounded): keypoints identical, orientations within budget, knife-edge descriptors a small share
    m = match_features(ref.features(), ref.descriptors(), fb, db)
    print("grid", {k: v for k, v in m.items() if k != "misses"})
    assert m["kp_miss"] == 0 and m["ori_miss"] <= budget(len(fb))["ori"]
    assert m["desc_miss"] <= 0.06 * m["desc_compared"] and m["max_desc_dist"] < 0.05, m
    ctx.close()


@pytest.mark.parametrize("name", [n for n in gu.cases() if n.startswith("mode_")])
def test_alternative_modes_match_reference_golden(capi, name):
    """HIP path vs fixtures produced by the reference's own kernels for the alternative modes."""
    g = gu.load(name)
    ctx = capi.Context(capi.default_config(**g["config"]))
    ctx.upload(g["image"])
    ctx.extract()
    for o in range(ctx.num_octaves):
        for l in range(ctx.num_levels):
            assert gu.sha1(ctx.dump_plane(capi.PLANE_GAUSS, o, l)) == g["plane_sha1"]["g_%d_%d" % (o, l)], (o, l)
    fb, db = ctx.download()
    fa, da = g["features"], g["descriptors"]
    assert len(fa) == len(fb) and len(da) == len(db)
    m = match_features(fa, da, fb, db, norm_scale=float(2 ** g["config"].get("norm_multi", 0)))
    print(name, {k: v for k, v in m.items() if k != "misses"})
    if g["config"].get("desc_mode", 0) == 2:
        # grid: knife-edge pixel snapping (test_grid_descriptor_mode).  Strict at stage level: the device's descriptors
        # against the oracle's for the device's own keypoint / orientation bits (the oracle's planes are the
        # fixture's, SHA-1 checked above; its grid stage is pinned on the fixture's keypoints in tests/test_golden_cpu.py)
        from oracle import pyoracle as po
        ro = po.run(po.default_config(**g["config"]), g["image"])
        assert_descriptor_rows(ro.describe(ctx.dump_extrema(), len(db)), db, len(fb), what="grid stage, fixture image")
        # end to end against the fixture: ~92 descriptors, the share of knife-edge ones is bounded loosely
        assert m["kp_miss"] == 0 and m["desc_miss"] <= 0.15 * max(1, m["desc_compared"]) and m["max_desc_dist"] < 0.05, m
    else:
        assert_parity(m, what=name, **budget(len(fa)))
    ctx.close()


def test_alt_descriptor_window_equals_plane():
    """k_descriptors_alt reads a window of the plane it staged in LDS; keypoints whose window exceeds 84 x 84 texels read the
    plane in HBM.  Both must give the same bits: every mode, windows over the image border, a configuration whose large
    keypoints take the plane path (the digest is over the sorted descriptor rows: the order of the features is not fixed)."""
    import json, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for win in ("1", "0"):
        p = subprocess.run([sys.executable, os.path.join(here, "alt_window_worker.py")], capture_output=True, text=True,
                           env=dict(os.environ, POPSIFT_ALT_WINDOW=win), timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[win] = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(res["1"]) == 12 and all(r["n"] > 50 for r in res["1"])
    for a, b in zip(res["1"], res["0"]):
        assert a == b, (a, b)


def test_fused_mode_kernels_equal_per_level_kernels():
    """Round 6: Fixed9 / Fixed15 run one kernel per octave (pyramid_fixed.hip) and VLFeat_Relative one fused kernel per level
    (pyramid_interp.hip, blur_interp.h).  Both must give the bits of the per-level kernels of pyramid_alt.hip, which the tests
    above hold to the oracle and the reference fixtures at small sizes: here at 1080p and at sizes whose upsampled planes cross
    2048 / 4096 columns (where the relative mode's fixed-point weights change with the coordinate's binade), byte and float
    input, every scaling variant.  POPSIFT_INTERP_LITERAL=1 forces the per-element (literal) weight path everywhere."""
    import json, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for tag, env in (("fused", {}), ("per_level", dict(POPSIFT_FIXED_FUSED="0", POPSIFT_INTERP_FUSED="0")),
                     ("literal", dict(POPSIFT_INTERP_LITERAL="1")), ("no_diagonal", dict(POPSIFT_INTERP_DIAGONAL="0"))):
        p = subprocess.run([sys.executable, os.path.join(here, "fused_modes_worker.py")], capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(res["fused"]) >= 19 and sum(r["n"] > 100 for r in res["fused"]) >= 10
    for tag in ("per_level", "literal", "no_diagonal"):
        for a, b in zip(res["fused"], res[tag]):
            assert a == b, (tag, a, b)
