"""GPU parity on adversarial image content (tests/adversarial.py): saturated 0 / 255 plateaus, a binary checkerboard,
text-like binary strokes, a sine grating, a pure ramp, bars with exact DoG ties, and a mosaic of all of them -- the
inputs that hit the strict 26-neighbour tests (s_extrema.cu:56-120: every exact tie must be rejected on both sides)
and the edge test (s_extrema.cu:491), which band-limited noise + blobs (popsift_amd/synth.py) never does.
Bars as everywhere: planes bit-exact, initial extrema identical, features within budget().  The same content as
reference-generated fixtures (tests/golden/ref_adv_*.npz) goes through test_hip_matches_reference_golden."""
import numpy as np
import pytest

from tests import adversarial as adv
from tests.parity import assert_parity, budget, match_features, sort_iext

pytestmark = pytest.mark.gpu

CONFIGS = [dict(octaves=5), dict(octaves=5, sift_mode=2), dict(octaves=4, sift_mode=1, gauss_mode=3)]


def _check(oracle, capi, img, kw, what):
    ref = oracle.run(oracle.default_config(**kw), img)
    ctx = capi.Context(capi.default_config(**kw))
    ctx.upload(img)
    ctx.extract()
    assert ctx.num_octaves == ref.num_octaves
    n_ext = 0
    for o in range(ref.num_octaves):
        assert ctx.octave_dims(o) == ref.dims[o]
        for l in range(ref.num_levels):
            g, r = ctx.dump_plane(capi.PLANE_GAUSS, o, l), ref.gauss(o, l)
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), "%s: plane (%d, %d) differs" % (what, o, l)
        a, b = sort_iext(ref.iext(o)), sort_iext(ctx.dump_iext(o))
        assert len(a) == len(b), "%s: octave %d has %d vs %d initial extrema" % (what, o, len(a), len(b))
        for f in ("xpos", "ypos", "lpos"):
            assert np.array_equal(a[f], b[f]), (what, o, f)
        n_ext += len(a)
    fb, db = ctx.download()
    fa, da = ref.features(), ref.descriptors()
    assert len(fa) == len(fb) == n_ext
    if len(fa):
        assert_parity(match_features(fa, da, fb, db), what=what + " oracle -> HIP", **budget(len(fa)))
        assert_parity(match_features(fb, db, fa, da), what=what + " HIP -> oracle", **budget(len(fa)))
    ctx.close()
    return n_ext


@pytest.mark.parametrize("name", sorted(adv.CONTENT))
def test_adversarial_content_640x480(oracle, capi, name):
    img = adv.make(name, 640, 480)
    counts = [_check(oracle, capi, img, kw, "%s %s" % (name, kw)) for kw in CONFIGS]
    if name in ("grating", "ramp"):
        assert counts == [0, 0, 0]              # ties / monotone: not one strict extremum
    else:
        assert max(counts) > 0


def test_adversarial_mosaic_1080p(oracle, capi):
    """The mosaic at the bench workload's size (octave 0 = 3840 x 2160): the large-plane kernels on hard content."""
    n = _check(oracle, capi, adv.make("composite", 1920, 1080), dict(octaves=5), "composite 1080p")
    assert n > 500


def test_adversarial_float_input_and_no_upsampling(oracle, capi):
    img = adv.make("composite", 400, 300).astype(np.float32) / 256.0
    ref_kw = dict(octaves=3, upscale_factor=0.0)
    ref = oracle.run(oracle.default_config(**ref_kw), img)
    ctx = capi.Context(capi.default_config(**ref_kw))
    ctx.upload(img)
    ctx.extract()
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            assert np.array_equal(ctx.dump_plane(capi.PLANE_GAUSS, o, l), ref.gauss(o, l)), (o, l)
        a, b = sort_iext(ref.iext(o)), sort_iext(ctx.dump_iext(o))
        assert len(a) == len(b) and np.array_equal(a["xpos"], b["xpos"]) and np.array_equal(a["ypos"], b["ypos"])
    fb, db = ctx.download()
    assert_parity(match_features(ref.features(), ref.descriptors(), fb, db), what="float mosaic", **budget(len(fb)))
    ctx.close()
