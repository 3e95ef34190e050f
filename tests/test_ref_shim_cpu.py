"""Live comparison of the oracle with the reference's own code on the CPU (oracle/_ref).

Only runs where oracle/_ref/libpopsift_ref.so has been built (needs /root/reference at build time;
the library travels with the snapshot, the sources do not).  Slow (fiber emulation of every CUDA
thread), so the images are tiny; the committed fixtures in tests/golden cover more cases.
"""
import numpy as np
import pytest

from popsift_amd.synth import synth
from tests.parity import match_features


@pytest.fixture(scope="module")
def ref():
    from oracle import pyref
    if not pyref.available():
        pytest.skip("oracle/_ref/libpopsift_ref.so not built (no /root/reference here)")
    pyref.lib()
    return pyref


@pytest.mark.parametrize("kw", [dict(), dict(gauss_mode=3), dict(upscale_factor=0.0), dict(levels=4, sigma=1.4)])
def test_gauss_tables_bit_equal_reference(oracle, ref, kw):
    a = ref.gauss_tables(oracle.default_config(**kw))
    b = oracle.gauss_tables(oracle.default_config(**kw))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_pipeline_matches_reference(oracle, ref):
    img = synth(88, 64, 31)
    cfg = oracle.default_config(octaves=3)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.dims == o.dims and r.num_levels == o.num_levels
    for oc in range(r.num_octaves):
        for l in range(r.num_levels):
            assert np.array_equal(r.gauss(oc, l), o.gauss(oc, l)), (oc, l)     # bit-identical planes
        for l in range(r.num_levels - 1):
            assert np.array_equal(r.dog(oc, l), o.dog(oc, l))
        assert len(r.iext(oc)) == len(o.iext(oc))
    m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors())
    assert m["kp_match"] == 1.0 and m["ori_match"] == 1.0 and m["desc_match"] == 1.0
    # the public API (PopSift::enqueue -> SiftJob::get, worker threads) gives the same counts
    ra = ref.run(cfg, img, api=True)
    assert (ra.ext_total, ra.ori_total) == (r.ext_total, r.ori_total)


@pytest.mark.parametrize("mode,grid,frac", [(0, 2, 3), (1, 3, 2), (2, 2, 5)])
def test_grid_filter_matches_reference(oracle, ref, mode, grid, frac):
    """extrema_filter_grid: the reference's own s_filtergrid.cu (on oracle/ref_shim/thrust_shim.h) vs the
    oracle's grid_filter().  The emulated device fills the extrema buffers in a deterministic order, so
    even RandomScale (buffer-order dependent) can be compared feature by feature here."""
    img = synth(160, 120, 99)
    total = oracle.run(oracle.default_config(octaves=3), img).ext_total
    cfg = oracle.default_config(octaves=3, filter_max_extrema=total // frac, filter_grid_size=grid,
                                grid_filter_mode=mode)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.ext_total == o.ext_total < total and r.ori_total == o.ori_total
    for oc in range(r.num_octaves):
        kept = o.iext(oc)
        assert len(r.iext(oc)) == int((kept["ignore"] == 0).sum())
    m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors())
    assert m["kp_match"] == 1.0 and m["ori_match"] == 1.0 and m["desc_match"] == 1.0, m


@pytest.mark.parametrize("nl,nr,seed", [(40, 50, 1), (3, 1, 2), (33, 70, 3)])
def test_matcher_matches_reference(oracle, ref, nl, nr, seed):
    """osift_match vs the reference's own FeaturesDev::match (features.cu on the CUDA emulation; its
    result is parsed from what show_distance prints): same indices, same accept flags, distances equal
    to the 3 printed decimals."""
    rng = np.random.default_rng(seed)
    left = rng.random((nl, 128), dtype=np.float32)
    right = rng.random((nr, 128), dtype=np.float32)
    if nr > 10:
        right[7] = left[3]; right[9] = left[3]           # duplicates: d1 == d2 == 0 -> NaN ratio -> reject
    mr, dr = ref.match(left, right)
    mo, do_ = oracle.match(left, right)
    assert np.array_equal(mr, mo)
    finite = np.isfinite(do_)
    assert np.abs(dr[finite] - do_[finite]).max() <= 6e-4


def _ref_fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        w, h = int(rng.integers(40, 110)), int(rng.integers(40, 90))
        kw = dict(octaves=int(rng.integers(1, 4)), levels=int(rng.integers(2, 5)), sift_mode=int(rng.integers(0, 3)),
                  gauss_mode=int(rng.choice([0, 3])), upscale_factor=float(rng.choice([-1.0, 0.0, 1.0])),
                  norm_mode=int(rng.integers(0, 2)), norm_multi=int(rng.choice([0, 9])),
                  sigma=float(rng.choice([1.2, 1.6, 2.0])), threshold=float(rng.choice([0.02, 0.04])))
        out.append((w, h, 7000 + i, kw))
    return out


@pytest.mark.parametrize("w,h,seed,kw", _ref_fuzz_cases(16, 4321))
def test_fuzz_oracle_vs_reference(oracle, ref, w, h, seed, kw):
    """Seeded random configurations through the reference's own code and the oracle: planes bit-identical,
    same initial extrema, identical feature sets."""
    img = synth(w, h, seed)
    cfg = oracle.default_config(**kw)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.dims == o.dims and r.num_levels == o.num_levels
    for oc in range(r.num_octaves):
        for l in range(r.num_levels):
            assert np.array_equal(r.gauss(oc, l), o.gauss(oc, l)), (oc, l)
        assert len(r.iext(oc)) == len(o.iext(oc))
    assert r.ext_total == o.ext_total and r.ori_total == o.ori_total
    if r.ext_total:
        m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors(), norm_scale=float(2 ** kw["norm_multi"]))
        assert m["kp_match"] == 1.0 and m["ori_match"] == 1.0 and m["desc_match"] == 1.0, m


@pytest.mark.parametrize("up,mode,is_float", [(2.0, 2, False), (-2.0, 0, False), (0.5, 1, False), (1.5, 2, True), (-0.5, 0, False)])
def test_other_scale_factors_match_reference(oracle, ref, up, mode, is_float):
    """setDownsampling values other than -1 / 0 / +1, fractional ones included (popsift.cpp:109-126: the octave-0 size
    is ceil(w * 2^up), the texture coordinates of s_pyramid_build.cu:96-126 follow): planes bit-identical, same extrema,
    identical feature sets."""
    from popsift_amd.synth import synth_float
    w, h = (200, 152) if up < -1 else (72, 56)
    _scale_factor_case(oracle, ref, w, h, up, mode, is_float)


@pytest.mark.parametrize("w,h,up,mode", [(163, 122, 0.5, 1), (201, 151, -0.5, 0)])
def test_fractional_scale_factors_on_the_rounding_boundaries(oracle, ref, w, h, up, mode):
    """Sizes at which some tap coordinates of octave 0 sit on a 1/256 sub-texel boundary, so that (x + shift)/W -+ k/W
    (the reference, s_pyramid_build_ra.cu:35-50) and (x -+ k + shift)/W give different bits
    (tests/test_oracle_cpu.py::test_literal_texture_form_is_what_fractional_scale_factors_get; these two sizes: 858 and 658
    pixels of level 0 differ between the forms, by up to 0.015): the oracle must follow the reference's form."""
    _scale_factor_case(oracle, ref, w, h, up, mode, False, octaves=2)


def _scale_factor_case(oracle, ref, w, h, up, mode, is_float, octaves=None):
    from popsift_amd.synth import synth_float
    img = synth_float(w, h, 11) if is_float else synth(w, h, 11)
    cfg = oracle.default_config(octaves=octaves or (3 if up > -2 else 2), upscale_factor=up, sift_mode=mode)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.dims == o.dims and r.num_levels == o.num_levels
    for oc in range(r.num_octaves):
        for l in range(r.num_levels):
            assert np.array_equal(r.gauss(oc, l), o.gauss(oc, l)), (oc, l)
        assert len(r.iext(oc)) == len(o.iext(oc))
    assert r.ext_total == o.ext_total and r.ori_total == o.ori_total
    if r.ext_total:
        m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors())
        assert m["kp_match"] == 1.0 and m["ori_match"] == 1.0 and m["desc_match"] == 1.0, m


# ---- alternative pyramid / descriptor modes --------------------------------------
@pytest.mark.parametrize("kw", [dict(gauss_mode=1), dict(gauss_mode=2), dict(gauss_mode=4), dict(gauss_mode=5),
                                dict(gauss_mode=1, levels=4, sigma=1.4, upscale_factor=0.0)])
def test_alternative_gauss_tables_bit_equal_reference(oracle, ref, kw):
    """abs_o0 / abs_oN / interpolated (ratio, multiplier) tables of init_filter (gauss_filter.cu:188-214, 373-410)."""
    a = ref.gauss_tables(oracle.default_config(**kw))
    b = oracle.gauss_tables(oracle.default_config(**kw))
    for k in a:
        # unused levels (sigma 0) of the fixed-span modes hold 0/0 in both
        assert np.array_equal(a[k], b[k], equal_nan=True), k


ALT_PYRAMIDS = [
    dict(gauss_mode=1),                                   # VLFeat_Relative: bilinear-paired taps
    dict(gauss_mode=2),                                   # VLFeat_Relative_All: octave 0 straight from the input
    dict(gauss_mode=4), dict(gauss_mode=5),               # Fixed9 / Fixed15
    dict(scaling_mode=0),                                 # ScaleDirect
    dict(scaling_mode=0, gauss_mode=1), dict(scaling_mode=0, gauss_mode=4), dict(scaling_mode=0, gauss_mode=3),
    dict(gauss_mode=1, upscale_factor=0.0, sift_mode=1), dict(gauss_mode=2, upscale_factor=-1.0, sift_mode=2),
]


@pytest.mark.parametrize("kw", ALT_PYRAMIDS)
def test_alternative_pyramid_modes_match_reference(oracle, ref, kw):
    """Every branch of Pyramid::build_pyramid (s_pyramid_build.cu:478-546) through the reference's own kernels and
    the oracle: planes bit-identical, same extrema, identical feature sets."""
    img = synth(96, 72, 5)
    cfg = oracle.default_config(octaves=3, **kw)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.dims == o.dims and r.num_levels == o.num_levels
    for oc in range(r.num_octaves):
        for l in range(r.num_levels):
            assert np.array_equal(r.gauss(oc, l), o.gauss(oc, l)), (oc, l)
        for l in range(r.num_levels - 1):
            assert np.array_equal(r.dog(oc, l), o.dog(oc, l)), (oc, l)
        assert len(r.iext(oc)) == len(o.iext(oc))
    assert r.ext_total == o.ext_total and r.ori_total == o.ori_total
    m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors())
    assert m["kp_miss"] == 0 and m["ori_miss"] == 0 and m["desc_miss"] == 0, m


def test_fixed_modes_need_three_levels(oracle, ref):
    """make_octave only exists for levels = 3 (s_pyramid_fixed.cu:270-292): the reference throws, the oracle refuses."""
    img = synth(64, 48, 1)
    cfg = oracle.default_config(octaves=2, gauss_mode=4, levels=4)
    with pytest.raises(RuntimeError):
        oracle.run(cfg, img)


@pytest.mark.parametrize("desc_mode,name", [(1, "iloop"), (3, "igrid"), (4, "notile")])
def test_interpolating_descriptor_modes_match_reference(oracle, ref, desc_mode, name):
    """ext_desc_iloop / igrid / notile: bilinear gradients in the keypoint frame.  Keypoint positions of oracle and
    reference differ by <= 8e-6 px (contraction choices inside solve), which moves some 1.8 fixed-point bilinear
    weights by one step: descriptors agree to ~5e-5 (tolerance 1e-3)."""
    img = synth(120, 90, 17)
    cfg = oracle.default_config(octaves=3, desc_mode=desc_mode)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.ext_total == o.ext_total and r.ori_total == o.ori_total > 40
    m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors())
    assert m["kp_miss"] == 0 and m["ori_miss"] == 0 and m["desc_miss"] == 0 and m["max_desc_dist"] < 2e-4, m
    # and the mode really is a different descriptor than "loop"
    loop = oracle.run(oracle.default_config(octaves=3), img)
    assert np.abs(loop.descriptors() - o.descriptors()).max() > 1e-2


def test_grid_descriptor_mode_matches_reference(oracle, ref):
    """ext_desc_grid snaps its 16 x 16 sample points per tile to pixels through  (int)(pt + (round(pt + pix) - pt))
    (s_desc_grid.cu:72-78): whenever |round(..)| < |pt| / 2 the float sum may land one ulp BELOW the integer and
    truncate to the neighbouring pixel -- decided by the last bits of pt, i.e. of the keypoint position and of
    sin / cos of the orientation.  The descriptor STAGE is therefore pinned strictly on the reference's own
    keypoints: the oracle redoes the descriptors for exactly the reference's positions and orientation bits
    (osift_describe; both evaluate sin / cos in double, rounded once) and every descriptor agrees within 1e-3.
    End to end, oracle and reference positions differ in the last bit for some keypoints (FMA contraction in
    solve()), which moves knife-edge samples of a share of the descriptors: bounded, not strict."""
    from tests.parity import assert_descriptor_rows, extrema_from_features
    img = synth(200, 150, 3)
    cfg = oracle.default_config(octaves=4, desc_mode=2)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.ext_total == o.ext_total and r.ori_total == o.ori_total > 150
    rf, rd = r.features(), r.descriptors()
    ext = extrema_from_features(rf, int(cfg.upscale_factor), o.features(), o.extrema()["lpos"])
    worst = assert_descriptor_rows(o.describe(ext, len(rd)), rd, len(rf), what="grid descriptor stage on the reference's keypoints")
    assert worst < 1e-3
    m = match_features(rf, rd, o.features(), o.descriptors())
    assert m["kp_miss"] == 0 and m["ori_miss"] == 0
    assert m["desc_miss"] <= 0.1 * m["desc_compared"] and m["max_desc_dist"] < 0.05, m


def test_fixed_span_tap_rows_are_one_fma(oracle, ref):
    """fixedSpan::relativeTexAddress::octave_fixed_vert reads its taps at "ypos -+ i * mul_h" (s_pyramid_fixed.cu:140-141),
    which nvcc contracts into one fma; with a multiply rounded on its own a tap of row 161 of this 193 x 191 plane (Fixed15,
    scale factor 0.5) lands on the other side of a 1/256 sub-texel boundary and the whole row differs by 6e-4 (found by
    tools/ref_fuzz.py, round 4).  Oracle = reference, every plane."""
    img = synth(136, 135, 9080)
    cfg = oracle.default_config(octaves=1, sift_mode=1, gauss_mode=5, levels=3, upscale_factor=0.5, initial_blur=0.8, threshold=0.02)
    r, o = ref.run(cfg, img), oracle.run(cfg, img)
    assert r.dims == o.dims == [(193, 191)]
    for l in range(r.num_levels):
        assert np.array_equal(r.gauss(0, l), o.gauss(0, l)), l
    assert r.ext_total == o.ext_total
