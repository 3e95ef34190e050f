"""GPU tests of the API surface added in round 2: measurement entry points, FeaturesDev feature records,
descriptor-buffer growth, the flat C binding of PopSift / SiftJob (include/popsift_c.h) and the per-PopSift
automatic octave count."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from tests.parity import assert_parity, budget, match_features

pytestmark = pytest.mark.gpu


def test_copy_bench_and_blur_probe(capi):
    gbs, ms = capi.copy_bench(0, 1 << 28, 3)          # 256 MiB + 256 MiB: quick; bench.py uses 1 GiB
    assert 500.0 < gbs < 8000.0 and ms > 0
    ctx = capi.Context(capi.default_config(octaves=3))
    ctx.upload(synth(1024, 768, 1))
    ctx.enable_blur_probe(True)
    ctx.extract()
    ctx.extract()
    ms_l, by = ctx.blur_probe_times()
    iso = [ctx.time_blur(0, l, 10)[0] for l in range(1, ctx.num_levels)]
    if len(ms_l) == 1:
        # k_pyramid_flow: ONE launch for every blur level of the frame; bytes = 8 B per pixel and blurred plane of every
        # octave (+ 4 B per decimated pixel)
        px = [2048 * 1536, 1024 * 768, 512 * 384]
        want = 8.0 * 5 * sum(px) + 4.0 * (px[1] + px[2])
        assert abs(by - want) < 1e-6 * want, (by, want)
        assert 0.0 < ms_l[0] < 5.0 and 0.2 < ms_l[0] / sum(iso) < 8.0, (ms_l, iso)
    else:
        # 8 B per octave-0 pixel; launches that also carry a level of octave 1 (a quarter of the pixels) count those too
        assert len(ms_l) == ctx.num_levels - 1 and 8.0 * 2048 * 1536 <= by <= 1.1001 * 8.0 * 2048 * 1536
        assert all(0.0 < m < 5.0 for m in ms_l), ms_l
        # same launches replayed in isolation: same order of magnitude
        assert all(0.2 < a / b < 5.0 for a, b in zip(ms_l, iso)), (ms_l, iso)
    # the probe does not change results
    f1, d1 = ctx.download()
    ctx.enable_blur_probe(False)
    ctx.extract()
    f2, d2 = ctx.download()
    assert len(f1) == len(f2) and len(d1) == len(d2)
    ctx.close()


def test_features_dev_records_point_into_descriptor_array(capi):
    """FeaturesDev::getFeatures() (features.h:104-122): device array of popsift::Feature (72 bytes) whose desc[]
    are device pointers into the cloned descriptor array, as Pyramid::clone_device_descriptors leaves them."""
    ctx = capi.Context(capi.default_config(octaves=3))
    ctx.upload(synth(400, 300, 9))
    ctx.extract()
    fh, dh = ctx.download()
    fd, dd, rev, base = ctx.clone_results()
    assert capi.FEATURE_DEV_DTYPE.itemsize == 72
    assert len(fd) == len(fh) and np.array_equal(dd, dh)
    for name in ("debug_octave", "xpos", "ypos", "sigma", "num_ori", "orientation"):
        assert np.array_equal(fd[name], fh[name]), name
    idx = fh["desc_idx"].astype(np.int64)
    expect = np.where(idx >= 0, base + idx * 512, 0).astype(np.uint64)
    assert np.array_equal(fd["desc"], expect)
    # reverse map: descriptor j belongs to the keypoint that lists it
    for k in range(4):
        sel = fh["num_ori"] > k
        assert np.array_equal(rev[fh["desc_idx"][sel, k]], np.flatnonzero(sel))
    ctx.close()


def test_descriptor_buffers_grow_like_realloc_extrema(oracle, capi):
    """max_extrema = 300 per octave: the descriptor buffers start at 2 x 300 entries (sift_pyramid.cu:154-159);
    the frame has ~1000 extrema > max_extrema, so the reference's reallocExtrema grows them to 2 x 2048.  The HIP
    path grows after the counter read-back and reruns scan + descriptors.  Which 300 extrema of a full octave
    survive is arrival order (atomicAdd) on both sides, so the result is checked as a subset of the uncapped
    oracle run: every keypoint has all its descriptors, each equal to the oracle's."""
    img = synth(640, 480, 77)
    kw = dict(octaves=4, max_extrema=300)
    ref = oracle.run(oracle.default_config(**kw), img)
    full = oracle.run(oracle.default_config(octaves=4), img)
    assert ref.ext_total > 600 and ref.ori_total > 600
    for export in (False, True):
        ctx = capi.Context(capi.default_config(**kw))
        if export:
            fbuf = np.zeros(4096 * capi.FEATURE_DTYPE.itemsize, np.uint8)
            dbuf = np.zeros(8192 * 128, np.float32)
            ctx.attach_export(fbuf, dbuf)
        ctx.upload(img)
        for frame in range(2):                         # the second frame runs on the grown buffers
            ctx.extract()
            ne, no = ctx.counts()
            fb, db = ctx.exported() if export else ctx.download()
            assert ne == ref.ext_total and no > 600
            assert no == int(fb["num_ori"].sum()) == len(db)
            idx = np.concatenate([fb["desc_idx"][fb["num_ori"] > k, k] for k in range(4)])
            assert np.array_equal(np.sort(idx), np.arange(no))         # no descriptor index dropped (-1) or repeated
            m = match_features(fb, db, full.features(), full.descriptors())
            assert_parity(m, what="regrow export=%s frame %d" % (export, frame), **budget(len(fb)))
        ctx.close()


def test_c_binding_matches_c_abi(oracle, capi):
    """include/popsift_c.h: PopSift::enqueue / SiftJob::get through the flat C binding gives the oracle's
    feature set (this is the path bench.py's headline number times)."""
    img = synth(512, 384, 31)
    ps = capi.PopSift(capi.default_config(octaves=4))
    jobs = [ps.enqueue(img) for _ in range(5)]
    res = [ps.get(j) for j in jobs]
    ps.close()
    ref = oracle.run(oracle.default_config(octaves=4), img)
    for fb, db in res:
        assert len(fb) == ref.ext_total
        assert_parity(match_features(ref.features(), ref.descriptors(), fb, db), what="C binding", **budget(len(fb)))


def test_auto_octaves_resolved_once_per_popsift(oracle, capi):
    """Config::octaves = -1: the octave count comes from the FIRST image a PopSift sees and then sticks
    (popsift.cpp:118-122), whichever of the 8 worker contexts extracts a later image."""
    small, large = synth(160, 120, 3), synth(640, 480, 4)
    first_oct = max(int(np.floor(np.log2(120.0)) - 3 + 2), 1)           # scaleFactor = 2 for the default x2 upsample
    ps = capi.PopSift(capi.default_config(octaves=-1))
    jobs = [ps.enqueue(small)] + [ps.enqueue(large) for _ in range(12)] + [ps.enqueue(small)]
    res = [ps.get(j) for j in jobs]
    ps.close()
    ref_l = oracle.run(oracle.default_config(octaves=first_oct), large)
    ref_s = oracle.run(oracle.default_config(octaves=first_oct), small)
    for (fb, db), ref in zip(res, [ref_s] + [ref_l] * 12 + [ref_s]):
        assert len(fb) == ref.ext_total, (len(fb), ref.ext_total)
        assert fb["debug_octave"].max() <= first_oct - 1
        assert_parity(match_features(ref.features(), ref.descriptors(), fb, db), what="auto octaves", **budget(len(fb)))


def _device_count(capi):
    import ctypes as C
    n = C.c_int()
    capi.lib().psx_device_count.argtypes = [C.POINTER(C.c_int)]
    assert capi.lib().psx_device_count(C.byref(n)) == 0
    return n.value


def test_second_device_when_present(oracle, capi):
    """PopSift(config, mode, imode, device) with device != 0 (popsift.h:158,166-168): the C-ABI context, the C++ pipeline
    and the per-device pinned pool on device 1, results against the oracle.  Skipped on a one-GPU box (the round-end
    8-GPU node runs it)."""
    if _device_count(capi) < 2:
        pytest.skip("one GPU visible")
    img = synth(640, 480, 77)
    ref = oracle.run(oracle.default_config(octaves=4), img)
    ctx = capi.Context(capi.default_config(octaves=4), device=1)
    ctx.upload(img)
    ctx.extract()
    fb, db = ctx.download()
    ctx.close()
    from tests.parity import assert_parity, budget, match_features
    assert len(fb) == ref.ext_total
    assert_parity(match_features(ref.features(), ref.descriptors(), fb, db), what="C-ABI on device 1", **budget(len(fb)))
    before = capi.pool_stats(1)
    ps = capi.PopSift(capi.default_config(octaves=4), device=1)
    res = [ps.get(ps.enqueue(img)) for _ in range(6)]
    ps.close()
    after = capi.pool_stats(1)
    assert after["hits"] + after["allocs"] > before["hits"] + before["allocs"], "device 1 must use ITS pinned pool"
    for f2, d2 in res:
        assert len(f2) == ref.ext_total
        assert_parity(match_features(ref.features(), ref.descriptors(), f2, d2), what="PopSift on device 1", **budget(len(f2)))


def test_zero_copy_export_of_two_contexts_on_two_streams_stays_separate(oracle, capi):
    """Two contexts (two HIP streams) exporting into their own pinned buffers while both are in flight
    (psx_attach_export: k_scan deposits the three counters, k_descriptors the records and descriptors, straight into
    mapped host memory).  Round 3's removed split-frame experiment saw a result that carried only its second part's
    orientation count: two scan kernels of ONE frame wrote the SAME counter words from two streams without an order
    between them (a missing cross-stream dependency of the experiment, not a scope problem of the stores).  What the
    product relies on is checked here: a context's export words are written by exactly one scan kernel per frame, on
    that context's stream, and the host reads them after that stream's frame event -- frames of different contexts
    never share a word.  300 alternating frames, every count and every exported array is its own context's."""
    imgs = [synth(512, 384, 11), synth(480, 360, 12)]
    refs = [oracle.run(oracle.default_config(octaves=4), im) for im in imgs]
    ctxs = [capi.Context(capi.default_config(octaves=4)) for _ in imgs]
    bufs = []
    for c in ctxs:
        pf = np.zeros(20000 * capi.FEATURE_DTYPE.itemsize, dtype=np.uint8)       # registered (pinned + mapped) by psx_attach_export
        pd = np.zeros(40000 * 128, dtype=np.float32)
        c.attach_export(pf, pd)
        bufs.append((pf, pd))
    for k, c in enumerate(ctxs):
        c.upload(imgs[k])
    for rep in range(150):
        for c in ctxs:
            c.extract()                                        # both frames in flight on their two streams
        for k, c in enumerate(ctxs):
            ne, no = c.counts()
            assert (ne, no) == (refs[k].ext_total, refs[k].ori_total), (rep, k, ne, no)
            if rep % 50 == 0:
                f, d = c.exported()
                from tests.parity import assert_parity, budget, match_features
                assert_parity(match_features(refs[k].features(), refs[k].descriptors(), f.copy(), d.copy()),
                              what="export of context %d, rep %d" % (k, rep), **budget(ne))
    for c in ctxs:
        c.attach_export(None, None)
        c.close()


def test_cross_stream_hand_over_of_plain_stores_under_pcie_load(capi):
    """The mechanism behind round 3's unexplained split-frame failure, isolated (util.hip psx_debug_cross_stream): a kernel
    busy with PCIe stores whose last workgroup writes a device word (plain store), an event, a reader kernel on a second
    stream (plain load), an event back -- 3000 rounds, with other contexts extracting on their own streams meanwhile.  Not
    one stale read: event order between streams carries plain stores on this stack, so the experiment lost its count to a
    missing edge in its own launch order (its two scan parts wrote the same export words), not to the scope of the export
    stores; the product never has two kernels of a frame on different streams."""
    import ctypes as C
    import threading
    L = capi.lib()
    L.psx_debug_cross_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    stop = threading.Event()

    def load():
        ctx = capi.Context(capi.default_config(octaves=4))
        ctx.upload(synth(1280, 720, 3))
        while not stop.is_set():
            ctx.extract(); ctx.counts()
        ctx.close()
    ts = [threading.Thread(target=load) for _ in range(3)]
    for t in ts:
        t.start()
    try:
        for words in (0, 8, 64):
            stale = C.c_int(-2)
            assert L.psx_debug_cross_stream(0, 1000, words, C.byref(stale)) == 0
            assert stale.value == 0, "stale reads across the event with %d PCIe words per thread: %d of 1000" % (words, stale.value)
    finally:
        stop.set()
        for t in ts:
            t.join()


_SWITCH_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import pyoracle as oracle
from popsift_amd import capi
from popsift_amd.synth import synth
from tests.parity import assert_parity, budget, match_features
for (w, h, seed, kw) in [(640, 480, 3, dict(octaves=4)), (333, 251, 5, dict(octaves=3, sift_mode=1)), (1920, 1080, 1000, dict(octaves=5, sift_mode=2))]:
    img = synth(w, h, seed)
    ref = oracle.run(oracle.default_config(**kw), img)
    ctx = capi.Context(capi.default_config(**kw)); ctx.upload(img); ctx.extract()
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            assert np.array_equal(ctx.dump_plane(capi.PLANE_GAUSS, o, l), ref.gauss(o, l)), (o, l)
    fb, db = ctx.download()
    assert len(fb) == ref.ext_total
    assert_parity(match_features(ref.features(), ref.descriptors(), fb, db), what="switch", **budget(len(fb)))
    ctx.close()
print("SWITCH-OK")
"""


@pytest.mark.parametrize("env", [dict(POPSIFT_DESC_DENORM="0"), dict(POPSIFT_LEVEL0_X2="0"), dict(POPSIFT_LEVEL0_FUSED="0"),
                                 dict(POPSIFT_BLUR_DEFER="0"), dict(POPSIFT_DESC_WGS="3"), dict(POPSIFT_BLUR_DMA="2"), dict(POPSIFT_BLUR_DMA="3"),
                                 dict(POPSIFT_BLUR_STEPS="3")])
def test_documented_fallback_switches_keep_parity(env):
    """The kernel-variant switches of INTEGRATION.md (read once per process, hence a subprocess each): the older variants
    they select stay bit-exact on the planes and within the feature budget on three frames."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT % root], env=e, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SWITCH-OK" in r.stdout, (env, r.stdout[-400:], r.stderr[-1500:])
