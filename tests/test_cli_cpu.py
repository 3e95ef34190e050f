"""CPU tests of the command line layer: the PGM/PPM reader, the option surface of
popsift-demo / popsift-match, and the CMake package (find_package(PopSift) + PopSift::popsift)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "popsift_amd", "lib")
APP = os.path.join(ROOT, "popsift_amd", "csrc", "host", "app")


@pytest.fixture(scope="module")
def built():
    from popsift_amd import build
    build.build_all()
    return LIB


@pytest.fixture(scope="module")
def pgm_tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pgm") / "test_pgmread")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", APP, os.path.join(ROOT, "tests", "cpp", "test_pgmread.cpp"),
                           os.path.join(APP, "pgmread.cpp"), "-o", exe])
    return exe


def _gray(rgb):
    r, g, b = (rgb[..., k].astype(np.uint32) for k in range(3))
    return ((4899 * r + 9617 * g + 1868 * b) >> 14).astype(np.uint8)          # pgmread.cpp:25-28


def _decode(tool, path, tmp_path):
    out = tmp_path / "decoded.raw"
    p = subprocess.run([tool, str(path), str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        return None
    w, h = (int(v) for v in p.stdout.split())
    return np.fromfile(str(out), np.uint8).reshape(h, w)


def test_pgm_ppm_reader_all_variants(pgm_tool, tmp_path):
    rng = np.random.default_rng(5)
    w, h = 37, 23
    g8 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    c8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    g16 = rng.integers(0, 1000, (h, w), dtype=np.uint16)
    c16 = rng.integers(0, 1000, (h, w, 3), dtype=np.uint16)
    scale = lambda v, m: (v.astype(np.float64) * 255.0 / m).astype(np.uint8)
    cases = {}
    # P5 / P6 binary, 8 bit, with comment lines between the header fields
    (tmp_path / "a.pgm").write_bytes(b"P5\n# a comment\n%d %d\n# another\n255\n" % (w, h) + g8.tobytes())
    cases["a.pgm"] = g8
    (tmp_path / "b.ppm").write_bytes(b"P6\n%d %d\n255\n" % (w, h) + c8.tobytes())
    cases["b.ppm"] = _gray(c8)
    # P2 / P3 ASCII, maxval 255 and maxval 999 (rescaled)
    (tmp_path / "c.pgm").write_text("P2\n%d %d\n255\n" % (w, h) + "\n".join(" ".join(str(v) for v in row) for row in g8) + "\n")
    cases["c.pgm"] = g8
    (tmp_path / "d.pgm").write_text("P2\n%d %d\n999\n" % (w, h) + " ".join(str(v) for v in g16.ravel()) + "\n")
    cases["d.pgm"] = scale(g16, 999)
    (tmp_path / "e.ppm").write_text("P3\n  %d %d\n255\n" % (w, h) + " ".join(str(v) for v in c8.ravel()) + "\n")
    cases["e.ppm"] = _gray(c8)
    # 16-bit binary: samples in host byte order, as the reference reads them (pgmread.cpp:177-189, 222-246)
    (tmp_path / "f.pgm").write_bytes(b"P5\n%d %d\n999\n" % (w, h) + g16.tobytes())
    cases["f.pgm"] = scale(g16, 999)
    (tmp_path / "g.ppm").write_bytes(b"P6\n%d %d\n999\n" % (w, h) + c16.tobytes())
    r, g, b = (c16[..., k].astype(np.uint32) for k in range(3))
    cases["g.ppm"] = ((4899 * r + 9617 * g + 1868 * b) >> 14).astype(np.uint8)
    for name, expect in cases.items():
        got = _decode(pgm_tool, tmp_path / name, tmp_path)
        assert got is not None, name
        assert got.shape == expect.shape and np.array_equal(got, expect), name
    # errors: wrong magic, truncated data, missing file
    (tmp_path / "x.pgm").write_bytes(b"P7\n3 3\n255\n" + bytes(9))
    (tmp_path / "y.pgm").write_bytes(b"P5\n30 30\n255\n" + bytes(10))
    (tmp_path / "z.pgm").write_bytes(b"P5\n-3 3\n255\n" + bytes(9))
    for name in ("x.pgm", "y.pgm", "z.pgm", "does_not_exist.pgm"):
        assert _decode(pgm_tool, tmp_path / name, tmp_path) is None, name


# every long option of the reference's popsift-demo (src/application/main.cpp:56-123)
DEMO_OPTIONS = ["help", "verbose", "log", "input-file", "octaves", "levels", "sigma", "threshold", "edge-threshold", "edge-limit",
                "downsampling", "initial-blur", "gauss-mode", "desc-mode", "popsift-mode", "vlfeat-mode", "opencv-mode",
                "direct-scaling", "norm-multi", "norm-mode", "root-sift", "filter-max-extrema", "filter-grid", "filter-sort",
                "print-gauss-tables", "print-dev-info", "print-time-info", "write-as-uchar", "dont-write", "pgmread-loading", "float-mode",
                "devices", "device-list"]          # the last two are additions (replicas in one process), not reference options


def test_demo_option_surface(built):
    p = subprocess.run([os.path.join(built, "popsift-demo"), "--help"], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "PopSift version" in p.stdout
    for o in DEMO_OPTIONS:
        assert "--" + o in p.stdout, o
    p = subprocess.run([os.path.join(built, "popsift-match"), "--help"], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0
    for o in ["left", "right", "octaves", "gauss-mode", "desc-mode", "filter-sort", "direct-scaling"]:
        assert "--" + o in p.stdout, o


def test_demo_rejects_bad_command_lines(built, tmp_path):
    demo = os.path.join(built, "popsift-demo")
    for argv in (["--no-such-option"], ["--octaves"], ["--octaves", "x", "-i", "a.pgm"], ["--gauss-mode", "nonsense", "-i", "a.pgm"], [],
                 ["--devices", "0", "-i", "a.pgm"], ["--device-list", "0,,1", "-i", "a.pgm"], ["--device-list", "gpu0", "-i", "a.pgm"]):
        p = subprocess.run([demo] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path))
        assert p.returncode != 0, argv
    # a missing input is "nothing to do" with a failure exit code (main.cpp:289-292)
    p = subprocess.run([demo, "-i", "missing.pgm"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path))
    assert p.returncode != 0 and "neither regular file nor directory" in p.stdout


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_package_consumer(built, tmp_path):
    """find_package(PopSift CONFIG) + target PopSift::popsift, as the reference's README documents for consumers
    (README.md:60-71): configure, build and run a ten-line consumer against the build-tree package."""
    src = os.path.join(ROOT, "tests", "cmake_consumer")
    b = str(tmp_path / "b")
    subprocess.check_call(["cmake", "-S", src, "-B", b, "-DPopSift_DIR=" + os.path.join(ROOT, "cmake")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", b], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(b, "consumer")], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "consumer linked against PopSift" in out.stdout
    # the install-tree package of the top-level CMakeLists.txt exports the same target name
    top = open(os.path.join(ROOT, "CMakeLists.txt")).read()
    assert "install(EXPORT PopSiftTargets NAMESPACE PopSift::" in top and "add_library(PopSift::popsift ALIAS popsift)" in top
