"""Octave / level counts that exercise the launch schedule of psx_build_pyramid: shared blur launches for octave pairs
that fit one round of workgroups, separate ones where they do not, batched extrema scans of the small octaves (more
than one batch at 10 octaves), octaves = 1 and 2 (no pair at all), 2..6 levels (the diagonal offset is `levels`).
Planes bit-exact, extrema identical, features within the budget."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from tests.parity import assert_parity, budget, match_features, sort_iext

pytestmark = pytest.mark.gpu

CASES = [
    (2048, 2048, dict(octaves=-1)),                      # 10 octaves: 3 large, 7 small (two extrema batches)
    (1500, 900, dict(octaves=-1)),
    (300, 200, dict(octaves=-1)),                        # every octave pair shares its launches
    (640, 480, dict(octaves=1)),
    (640, 480, dict(octaves=2)),
    (1024, 768, dict(octaves=7, levels=2)),
    (1024, 768, dict(octaves=7, levels=4)),
    (1024, 768, dict(octaves=7, levels=6)),
    (1920, 1080, dict(octaves=8, upscale_factor=0.0)),
    (1920, 1080, dict(octaves=6, upscale_factor=-1.0)),
    (4000, 300, dict(octaves=-1)),
    (300, 4000, dict(octaves=-1)),
]


@pytest.mark.parametrize("w,h,kw", CASES)
def test_launch_schedule_shapes(oracle, capi, w, h, kw):
    img = synth(w, h, 77)
    ref = oracle.run(oracle.default_config(**kw), img)
    ctx = capi.Context(capi.default_config(**kw))
    ctx.upload(img)
    ctx.extract()
    assert ctx.num_octaves == ref.num_octaves
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            assert np.array_equal(ctx.dump_plane(capi.PLANE_GAUSS, o, l), ref.gauss(o, l)), (o, l)
        a, b = sort_iext(ref.iext(o)), sort_iext(ctx.dump_iext(o))
        assert len(a) == len(b) and np.array_equal(a["xpos"], b["xpos"]) and np.array_equal(a["ypos"], b["ypos"]), o
    fb, db = ctx.download()
    fa, da = ref.features(), ref.descriptors()
    assert len(fa) == len(fb)
    if len(fa):
        assert_parity(match_features(fa, da, fb, db), what="%dx%d %s" % (w, h, kw), **budget(len(fa)))
    ctx.close()
