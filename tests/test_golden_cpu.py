"""The oracle against the reference's golden vectors (tests/golden/ref_*.npz).

The fixtures were produced by the reference's OWN sources (compiled for the CPU through
oracle/ref_shim); this is what pins the restatement in oracle/sift_oracle.c:
  * every Gaussian plane bit-identical (SHA-1 of the float32 plane),
  * initial extrema: same count, same level, positions within 2e-5 px (the reference's device code
    is FMA-contracted by nvcc, the oracle fixes one contraction-free order),
  * features / descriptors within the north_star tolerances, as set-match fractions of 1.0.
"""
import numpy as np
import pytest

from tests import golden_util as gu
from tests.parity import match_features, sort_iext


@pytest.mark.parametrize("name", gu.cases())
def test_oracle_matches_reference_golden(oracle, name):
    g = gu.load(name)
    cfg = oracle.default_config(**g["config"])
    r = oracle.run(cfg, g["image"])
    assert r.dims == g["dims"] and r.num_levels == g["num_levels"]
    for o in range(r.num_octaves):
        for l in range(r.num_levels):
            assert gu.sha1(r.gauss(o, l)) == g["plane_sha1"]["g_%d_%d" % (o, l)], (o, l)
    assert np.array_equal(r.gauss(r.num_octaves - 1, r.num_levels - 1), g["gauss_last"])
    for o in range(r.num_octaves):
        b = r.iext(o)
        a, b = sort_iext(g["iext_%d" % o]), sort_iext(b[b["ignore"] == 0])
        assert len(a) == len(b)
        if len(a):
            assert np.array_equal(a["lpos"], b["lpos"])
            # the reference's device code is FMA-contracted inside solve(): positions agree to 2e-5 px or 2 ulp
            for f in ("xpos", "ypos"):
                assert np.all(np.abs(a[f] - b[f]) <= np.maximum(2e-5, 2 * np.spacing(np.abs(a[f])))), f
            assert np.allclose(a["sigma"], b["sigma"], rtol=1e-6)
    fa, da = g["features"], g["descriptors"]
    fb, db = r.features(), r.descriptors()
    assert len(fa) == len(fb) and len(da) == len(db)
    scale = float(2 ** g["config"].get("norm_multi", 0))
    m = match_features(fa, da, fb, db, norm_scale=scale)
    assert m["kp_miss"] == 0 and m["ori_miss"] == 0, m
    if g["config"].get("desc_mode", 0) == 2:
        # grid descriptor: knife-edge pixel snapping (tests/test_ref_shim_cpu.py::test_grid_descriptor_mode_matches_reference).
        # Strict on the descriptor stage: the oracle redoes the descriptors for the fixture's own keypoint and
        # orientation bits (fixture regenerated in round 3 with the single-rounded sin / cos on both sides) ...
        from tests.parity import assert_descriptor_rows, extrema_from_features
        ext = extrema_from_features(fa, int(cfg.upscale_factor), fb, r.extrema()["lpos"])
        assert_descriptor_rows(r.describe(ext, len(da)), da, len(fa), what="grid stage on the fixture's keypoints", norm_scale=scale)
        # ... and bounded end to end, where last-bit position differences move knife-edge samples
        assert m["desc_miss"] <= 0.1 * max(1, m["desc_compared"]) and m["max_desc_dist"] < 0.05, m
    else:
        assert m["desc_miss"] == 0 and m["max_desc_dist"] < (2e-4 if g["config"].get("desc_mode", 0) else 1e-4), m


def test_golden_fixtures_cover_all_modes():
    names = gu.cases()
    assert len(names) >= 5
    modes = {gu.load(n)["config"].get("sift_mode", 0) for n in names}
    assert modes == {0, 1, 2}
