"""GPU tests of the command line tools: popsift-demo reads PGM/PPM, writes output-features.txt in
the reference's format (features.cu:310-330) and, with --log, the dir-octave / dir-dog / dir-desc dumps the reference's
regression protocol compares (testScripts/testOxfordDataset.sh.in:65-154); popsift-match prints the matcher's lines."""
import os
import subprocess

import numpy as np
import pytest

from popsift_amd.synth import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "popsift_amd", "lib", "popsift-demo")
MATCH = os.path.join(ROOT, "popsift_amd", "lib", "popsift-match")


def _write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes())


def _run(cmd, cwd):
    p = subprocess.run(cmd, cwd=str(cwd), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return p


def _expected_rows(ref):
    """One row per (keypoint, orientation): x y 1/s^2 0 1/s^2 d0..d127 (Feature::print, features.cu:310-330)."""
    fa, da = ref.features(), ref.descriptors()
    rows = []
    for f in fa:
        for k in range(f["num_ori"]):
            s = 1.0 / (float(f["sigma"]) ** 2)
            rows.append(np.concatenate([[f["xpos"], f["ypos"], s, 0.0, s], da[f["desc_idx"][k]]]))
    return np.array(rows)


def _sorted(rows):
    return rows[np.lexsort((rows[:, 5], rows[:, 2], rows[:, 1], rows[:, 0]))]


def test_demo_writes_reference_feature_format(oracle, tmp_path):
    img = synth(320, 240, 66)
    _write_pgm(tmp_path / "in.pgm", img)
    ref = oracle.run(oracle.default_config(octaves=4, sift_mode=2, norm_multi=9), img)
    exp = _sorted(_expected_rows(ref))
    # float descriptors (3 significant digits)
    p = _run([DEMO, "-i", "in.pgm", "--octaves", "4", "--vlfeat-mode", "--norm-multi=9", "--pgmread-loading"], tmp_path)
    assert "Number of feature points: %d number of feature descriptors: %d" % (ref.ext_total, ref.ori_total) in p.stderr
    got = _sorted(np.loadtxt(str(tmp_path / "output-features.txt"), ndmin=2))
    assert got.shape == exp.shape == (ref.ori_total, 5 + 128)
    assert np.all(got[:, 3] == 0.0)
    assert np.abs(got[:, :2] - exp[:, :2]).max() <= 2e-3          # default 6 significant digits of the stream
    assert np.allclose(got[:, 2], exp[:, 2], rtol=2e-5) and np.array_equal(got[:, 2], got[:, 4])
    assert np.abs(got[:, 5:] - exp[:, 5:]).max() <= 0.5 + 2e-3 * 512  # setprecision(3) on values up to 512
    # descriptors rounded to integers
    _run([DEMO, "--input-file=in.pgm", "--octaves=4", "--vlfeat-mode", "--norm-multi", "9", "--write-as-uchar"], tmp_path)
    got = _sorted(np.loadtxt(str(tmp_path / "output-features.txt"), ndmin=2))
    assert np.array_equal(got[:, 5:], np.round(got[:, 5:]))
    assert np.abs(got[:, 5:] - exp[:, 5:]).max() <= 0.5 + 0.51      # roundf of values within 1e-3 * 512
    # --dont-write leaves the file alone
    os.remove(str(tmp_path / "output-features.txt"))
    _run([DEMO, "-i", "in.pgm", "--octaves", "4", "--dont-write"], tmp_path)
    assert not os.path.exists(str(tmp_path / "output-features.txt"))


def test_demo_float_mode_and_directory_input(oracle, tmp_path):
    d = tmp_path / "imgs" / "sub"
    d.mkdir(parents=True)
    a, b = synth(200, 150, 1), synth(240, 160, 2)
    _write_pgm(tmp_path / "imgs" / "a.pgm", a)
    _write_pgm(d / "b.pgm", b)
    p = _run([DEMO, "-i", "imgs", "--octaves", "3", "--float-mode"], tmp_path)
    # float mode feeds v / 256 (main.cpp:241-245), not v / 255
    counts = []
    for im in (a, b):
        r = oracle.run(oracle.default_config(octaves=3), (im.astype(np.float32) / np.float32(256.0)).astype(np.float32))
        counts.append("Number of feature points: %d number of feature descriptors: %d" % (r.ext_total, r.ori_total))
    assert [l for l in p.stderr.splitlines() if l.startswith("Number of feature points")] == counts
    assert p.stdout.count("Loading ") == 2


def _read_p2(path):
    tok = open(path).read().split()
    assert tok[0] == "P2" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w)


def _read_dump(path):
    with open(path, "rb") as f:
        assert f.readline() == b"floats\n"
        w, h = (int(v) for v in f.readline().split())
        return np.frombuffer(f.read(), np.float32).reshape(h, w)


def test_demo_log_dumps_match_the_reference_protocol(oracle, tmp_path):
    """--log: the byte-comparable debug files.  Gaussian planes are bit-identical to the oracle's, so every derived file
    must be EXACTLY what the reference's writers (write_plane_2d.cu:50-175) produce from the oracle's planes."""
    img = synth(160, 120, 12)
    _write_pgm(tmp_path / "in.pgm", img)
    _run([DEMO, "-i", "in.pgm", "--octaves", "3", "--log"], tmp_path)
    ref = oracle.run(oracle.default_config(octaves=3), img)
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            g = ref.gauss(o, l)
            assert np.array_equal(_read_p2(str(tmp_path / "dir-octave" / ("pyramid-o-%d-l-%d.pgm" % (o, l)))), g.astype(np.int64))
            assert np.array_equal(_read_dump(str(tmp_path / "dir-octave-dump" / ("pyramid-o-%d-l-%d.dump" % (o, l)))), g)
        for l in range(ref.num_levels - 1):
            d = ref.dog(o, l)
            assert np.array_equal(_read_dump(str(tmp_path / "dir-dog-dump" / ("d-pyramid-o-%d-l-%d.dump" % (o, l)))), d)
            assert np.array_equal(_read_p2(str(tmp_path / "dir-dog-txt" / ("d-pyramid-o-%d-l-%d.txt" % (o, l)))), d.astype(np.int64) + 127)
            mn, mx = np.float32(d.min()), np.float32(d.max())
            scaled = ((d - mn) * (np.float32(255.0) / (mx - mn))).astype(np.uint8).astype(np.int64)
            assert np.array_equal(_read_p2(str(tmp_path / "dir-dog" / ("d-pyramid-o-%d-l-%d.pgm" % (o, l)))), scaled)
    desc = np.loadtxt(str(tmp_path / "dir-desc" / "desc-pyramid.txt"), ndmin=2)
    fpt = np.loadtxt(str(tmp_path / "dir-fpt" / "desc-pyramid.txt"), ndmin=2)
    assert desc.shape == (ref.ori_total, 4 + 128) and fpt.shape == (ref.ori_total, 4)
    assert np.all((fpt[:, 3] >= 0) & (fpt[:, 3] < 360.0001))


def test_match_tool(oracle, tmp_path):
    base = synth(336, 256, 77)
    a = np.ascontiguousarray(base[8:248, 8:328])
    b = np.ascontiguousarray(base[5:245, 3:323])
    _write_pgm(tmp_path / "l.pgm", a)
    _write_pgm(tmp_path / "r.pgm", b)
    p = _run([MATCH, "-l", "l.pgm", "-r", "r.pgm", "--octaves", "3"], tmp_path)
    ra, rb = oracle.run(oracle.default_config(octaves=3), a), oracle.run(oracle.default_config(octaves=3), b)
    assert "Number of descriptors: %d" % ra.ori_total in p.stdout and "Number of descriptors: %d" % rb.ori_total in p.stdout
    lines = [l for l in p.stdout.splitlines() if l.startswith(("accept", "reject"))]
    assert len(lines) == ra.ori_total
    assert sum(l.startswith("accept") for l in lines) > 0.3 * len(lines)


def test_demo_device_list_two_replicas_on_one_gpu(oracle, tmp_path):
    """popsift-demo --device-list 0,0 / --devices 1: one PopSift replica per listed device in ONE process (the
    reference's multi-GPU model, popsift.h:158,166-168), image i -> replica i mod N, results read in input order.
    Two replicas on device 0 (the reference cannot do that: global device symbols) give every image's oracle counts;
    the last image's output-features.txt equals the single-replica run's."""
    d = tmp_path / "imgs"
    d.mkdir()
    imgs = [synth(200 + 16 * i, 150 + 8 * i, 40 + i) for i in range(5)]
    for i, im in enumerate(imgs):
        _write_pgm(d / ("%02d.pgm" % i), im)
    want = []
    for im in imgs:
        r = oracle.run(oracle.default_config(octaves=3), im)
        want.append("Number of feature points: %d number of feature descriptors: %d" % (r.ext_total, r.ori_total))
    outs = {}
    for name, extra in (("two", ["--device-list", "0,0"]), ("one", ["--devices", "1"])):
        p = _run([DEMO, "-i", "imgs", "--octaves", "3"] + extra, tmp_path)
        got = [ln for ln in p.stderr.splitlines() if ln.startswith("Number of feature points")]
        assert got == want, (name, got, want)
        outs[name] = _sorted(np.loadtxt(str(tmp_path / "output-features.txt"), ndmin=2))
    assert outs["one"].shape == outs["two"].shape and np.allclose(outs["one"], outs["two"], atol=2e-3)
    bad = subprocess.run([DEMO, "-i", "imgs", "--device-list", "0,x"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert bad.returncode != 0 and "device-list" in bad.stderr


@pytest.mark.parametrize("mode,desc_mode", [("loop", 0), ("grid", 2), ("igrid", 3), ("notile", 4)])
def test_demo_runs_the_reference_test_script(oracle, tmp_path, mode, desc_mode):
    """testScripts/TEST.sh.in:8-27 of the reference: popsift-demo with --log --gauss-mode=relative, the grid filter
    (--filter-max-extrema=2000 --filter-grid=2 --filter-sort=down), --popsift-mode --octaves=8 --threshold=0.04
    --edge-threshold=10.0 --initial-blur=0.5 and, per DescMode, --write-as-uchar --norm-multi=9; the script then sorts
    output-features.txt.  Same command line here; the sorted file against the oracle with the same Config (the filter is
    active: the frame has more than 1.1 x 2000 keypoints)."""
    img = synth(800, 600, 4242)
    _write_pgm(tmp_path / "img3.pgm", img)
    cfg = dict(octaves=8, gauss_mode=1, sift_mode=0, desc_mode=desc_mode, norm_multi=9, threshold=0.04, edge_limit=10.0,
               assume_initial_blur=1, initial_blur=0.5, filter_max_extrema=2000, filter_grid_size=2, grid_filter_mode=1)
    ref = oracle.run(oracle.default_config(**cfg), img)
    unfiltered = oracle.run(oracle.default_config(**dict(cfg, filter_max_extrema=-1)), img)
    assert unfiltered.ext_total > 2200 and ref.ext_total < unfiltered.ext_total, "the frame must trigger the filter"
    p = _run([DEMO, "--log", "--gauss-mode=relative", "--filter-max-extrema=2000", "--filter-grid=2", "--filter-sort=down",
              "--popsift-mode", "--octaves=8", "--threshold=0.04", "--edge-threshold=10.0", "--initial-blur=0.5",
              "--desc-mode=%s" % mode, "--write-as-uchar", "--norm-multi=9", "-i", "img3.pgm"], tmp_path)
    assert "Number of feature points: %d number of feature descriptors: %d" % (ref.ext_total, ref.ori_total) in p.stderr
    got = np.loadtxt(str(tmp_path / "output-features.txt"), ndmin=2)
    exp = _expected_rows(ref)
    assert got.shape == exp.shape == (ref.ori_total, 5 + 128)
    assert np.array_equal(got[:, 5:], np.round(got[:, 5:]))
    # pair the rows by position and scale (the file holds 6 significant digits: sorting both sides does not line up
    # keypoints whose x agree to 1e-4), the orientations of one keypoint by their descriptor
    used = np.zeros(len(exp), bool)
    bad = 0
    for g in got:
        c = np.flatnonzero((np.abs(exp[:, 0] - g[0]) <= 2e-3) & (np.abs(exp[:, 1] - g[1]) <= 2e-3) &
                           (np.abs(exp[:, 2] - g[2]) <= 2e-5 * g[2] + 1e-9) & ~used)
        assert len(c) > 0, "no oracle row for %s" % g[:5]
        dd = np.abs(exp[c, 5:] - g[5:]).max(axis=1)
        used[c[dd.argmin()]] = True
        bad += dd.min() > 0.5 + 0.51                     # roundf of values within 1e-3 * 512
    assert used.all()
    # grid snaps samples to pixels on the last bits of the orientation: end to end a small share of knife-edge
    # descriptors moves (tests/test_gpu_modes.py::test_grid_descriptor_mode holds the descriptor stage strictly)
    allowed = int(0.06 * len(got)) if mode == "grid" else max(1, len(got) // 2000)
    assert bad <= allowed, "%d of %d descriptor rows differ" % (bad, len(got))
    assert os.path.isdir(str(tmp_path / "dir-octave")) and os.path.isdir(str(tmp_path / "dir-desc"))


def test_demo_runs_the_oxford_regression_command_line(oracle, tmp_path):
    """testScripts/testOxfordDataset.sh.in:48 of the reference: popsift-demo --log --gauss-mode vlfeat --desc-mode loop
    --popsift-mode --root-sift --downsampling -1 on a colour PPM (the Oxford images are PPMs); the script byte-compares the
    quantised Gaussian planes and the sorted feature files with stored ones.  Same command line on a synthetic PPM: the
    plane files from the oracle's planes (bit-identical pyramid), the feature rows against the oracle's."""
    g = synth(400, 320, 31)
    rgb = np.stack([g, np.roll(g, 3, axis=1), np.roll(g, 5, axis=0)], axis=2).astype(np.uint8)
    with open(str(tmp_path / "img1.ppm"), "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]) + rgb.tobytes())
    r, gg, b = (rgb[:, :, k].astype(np.int64) for k in range(3))
    gray = ((4899 * r + 9617 * gg + 1868 * b) >> 14).astype(np.uint8)          # pgmread.cpp: the reference's integer luma
    p = _run([DEMO, "--log", "--gauss-mode", "vlfeat", "--desc-mode", "loop", "--popsift-mode", "--root-sift",
              "--downsampling", "-1", "-i", "img1.ppm"], tmp_path)
    ref = oracle.run(oracle.default_config(gauss_mode=0, desc_mode=0, sift_mode=0, norm_mode=0, upscale_factor=1.0), gray)
    assert "Number of feature points: %d number of feature descriptors: %d" % (ref.ext_total, ref.ori_total) in p.stderr
    for o in range(ref.num_octaves):
        for l in range(ref.num_levels):
            assert np.array_equal(_read_p2(str(tmp_path / "dir-octave" / ("pyramid-o-%d-l-%d.pgm" % (o, l)))),
                                  ref.gauss(o, l).astype(np.int64)), (o, l)
    got = np.loadtxt(str(tmp_path / "output-features.txt"), ndmin=2)
    exp = _expected_rows(ref)
    assert got.shape == exp.shape == (ref.ori_total, 5 + 128) and ref.ori_total > 500
    used = np.zeros(len(exp), bool)
    for row in got:
        c = np.flatnonzero((np.abs(exp[:, 0] - row[0]) <= 2e-3) & (np.abs(exp[:, 1] - row[1]) <= 2e-3) &
                           (np.abs(exp[:, 2] - row[2]) <= 2e-5 * row[2] + 1e-9) & ~used)
        assert len(c) > 0, row[:5]
        dd = np.abs(exp[c, 5:] - row[5:]).max(axis=1)
        assert dd.min() <= 1e-3 + 0.6e-3, dd.min()                 # descriptors within 1e-3, 3 significant digits of values below 1
        used[c[dd.argmin()]] = True
    assert used.all()
