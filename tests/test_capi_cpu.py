"""CPU tests of the product's host logic: the C-ABI library loads and exports every symbol
include/popsift_hip.h declares, its device-free entry points agree with the oracle, and the
product fails loudly (never falls back to a CPU path) without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "popsift_hip.h")).read()
    declared = set(re.findall(r"\b(psx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(capi.SYMBOLS) == declared


def test_host_library_exports_the_c_binding(capi):
    """include/popsift_c.h (flat C binding of PopSift / SiftJob / FeaturesHost) vs libpopsift.so."""
    hdr = open(os.path.join(ROOT, "include", "popsift_c.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(popsift_c_[a-z0-9_]+)\s*\(", code))
    assert len(declared) >= 10
    H = capi.host_lib()
    assert not [s for s in sorted(declared) if not hasattr(H, s)]
    assert set(capi.HOST_SYMBOLS) == declared
    for forbidden in ("hipStream_t", "#include <hip", "torch", "std::"):
        assert forbidden not in code


def test_header_cites_reference_and_has_no_device_types():
    hdr = open(os.path.join(ROOT, "include", "popsift_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)      # strip comments
    for forbidden in ("hipStream_t", "hipError_t", "hipEvent_t", "#include <hip", "torch", "std::", "at::Tensor"):
        assert forbidden not in code, forbidden
    assert hdr.count(".cu:") + hdr.count(".cpp:") + hdr.count(".h:") > 25      # file:line citations


def test_config_defaults_match_oracle(capi, oracle):
    a, b = capi.default_config(), oracle.default_config()
    for name, _ in a._fields_:
        if hasattr(b, name):
            assert getattr(a, name) == getattr(b, name), name
    assert a.scaling_mode == capi.SCALE_DEFAULT and a.desc_mode == capi.DESC_LOOP
    assert abs(capi.lib().psx_peak_threshold(C.byref(a)) - oracle.lib().osift_peak_threshold(C.byref(b))) == 0


@pytest.mark.parametrize("kw", [dict(), dict(gauss_mode=3), dict(upscale_factor=0.0), dict(upscale_factor=-1.0),
                                dict(levels=5, sigma=1.2), dict(levels=2, sigma=2.0), dict(assume_initial_blur=0)])
def test_gauss_tables_bit_equal_oracle(capi, oracle, kw):
    a = capi.gauss_tables(capi.default_config(**kw))
    b = oracle.gauss_tables(oracle.default_config(**kw))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_gauss_table_errors(capi):
    with pytest.raises(capi.PopSiftError):
        capi.gauss_tables(capi.default_config(sigma=2.5))      # gauss_filter.cu:131-137
    with pytest.raises(capi.PopSiftError):
        capi.gauss_tables(capi.default_config(levels=10))      # gauss_filter.cu:138-144


def test_invalid_modes_are_rejected_before_touching_a_device(capi):
    """Enum values outside popsift::Config's, and the one combination the reference itself refuses (Fixed9 / Fixed15
    need levels = 3, s_pyramid_fixed.cu:270-292), fail in psx_create with the reference's message -- not with a
    device error."""
    for kw, msg in ((dict(desc_mode=7), "not yet"), (dict(gauss_mode=9), "Gauss filter"), (dict(scaling_mode=5), "scaling"),
                    (dict(gauss_mode=capi.GAUSS_FIXED9, levels=4), "Unsupported number of levels"),
                    (dict(sift_mode=3), "sift mode")):
        with pytest.raises(capi.PopSiftError) as e:
            capi.Context(capi.default_config(**kw))
        assert msg in str(e.value), (kw, str(e.value))


def test_no_cpu_fallback_without_gpu(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PopSiftError) as e:
        capi.Context(capi.default_config())
    assert "hipSetDevice" in str(e.value) or "device" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "popsift_amd")):
        if os.sep + "build" in base or os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".c")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt) and "oracle/sift_oracle.c" not in txt:
                    bad.append(os.path.join(base, f))
                elif "liboracle" in txt or "pyoracle" in txt:
                    bad.append(os.path.join(base, f))
    hdr = open(os.path.join(ROOT, "include", "popsift_hip.h")).read()
    assert "oracle" not in hdr
    assert not bad, bad
    # bench.py: the oracle is imported in exactly three places -- the CPU baseline, the post-run parity check and the
    # config5 leg's checker -- and none sits inside a timed leg (e2e_parity is a backend method called after timed();
    # config5_leg and extras() run last, outside every timed region)
    btxt = open(os.path.join(ROOT, "bench.py")).read()
    assert btxt.count("from oracle import pyoracle") == 3
    assert "from oracle import pyoracle as po" in btxt.split("def config5_leg")[1].split("def match_leg")[0]
    assert "def e2e_parity" in btxt and "from oracle import pyoracle as po" in btxt.split("def e2e_parity")[1].split("def ")[0]
    timed_body = btxt.split("def timed(step, drain, probe=None):")[1].split("def reduce_sum_dict")[0]
    assert "oracle" not in timed_body and "parity" not in timed_body
    # the step / drain closures the timed region runs: nothing but enqueue / get
    steps_body = btxt.split("def e2e_step(n=None):")[1].split("desc_probe =")[0]
    assert "oracle" not in steps_body and "parity" not in steps_body


def test_bench_frame_rule_is_round_robin_over_gpus():
    """bench.py's dispatch rule (BASELINE config 4): global frame i goes to GPU i mod N; every rank gets BATCH
    distinct base frames and the union over ranks is 8N distinct frames."""
    import bench
    for world in (1, 2, 4, 8):
        seeds = [[bench.frame_seed(j, r, world) for j in range(bench.BATCH)] for r in range(world)]
        flat = sorted(s for row in seeds for s in row)
        assert flat == list(range(1000, 1000 + bench.BATCH * world))
        for r, row in enumerate(seeds):
            assert all((s - 1000) % world == r for s in row)


def _host_lib():
    p = os.path.join(ROOT, "popsift_amd", "lib", "libpopsift.so")
    if not os.path.exists(p):
        from popsift_amd import build
        build.build_all()
    return p


def test_cpp_host_library_api(tmp_path):
    """Compile and run tests/cpp/test_host_api.cpp against libpopsift.so: Config surface, error
    convention, and that a job is always fulfilled (get() throws instead of hanging) with no GPU."""
    _host_lib()
    exe = str(tmp_path / "test_host_api")
    inc = [os.path.join(ROOT, "popsift_amd", "csrc", "include"), os.path.join(ROOT, "include")]
    libdir = os.path.join(ROOT, "popsift_amd", "lib")
    cmd = ["g++", "-std=c++14", "-O1", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"), "-o", exe]
    for i in inc:
        cmd += ["-I", i]
    cmd += ["-L", libdir, "-lpopsift", "-lpopsift_hip", "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    assert "ALL OK" in out.stdout


def test_cpp_host_pools(tmp_path):
    """Compile and run tests/cpp/test_host_pool.cpp (internal header host_pool.h) against libpopsift.so: the size-ordered
    free list returns the smallest adequate buffer, bounds how oversized a reused buffer may be, survives 8 threads, and the
    NUMA helper leaves the affinity alone for devices it cannot resolve (no GPU involved: pageable pool only)."""
    _host_lib()
    exe = str(tmp_path / "test_host_pool")
    libdir = os.path.join(ROOT, "popsift_amd", "lib")
    cmd = ["g++", "-std=c++14", "-O1", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_host_pool.cpp"), "-o", exe,
           "-I", os.path.join(ROOT, "popsift_amd", "csrc", "host"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "popsift_amd", "csrc", "include"), "-L", libdir, "-lpopsift", "-lpopsift_hip", "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout


def test_bench_algorithmic_bytes_match_the_survey():
    """bench.py's roofline numerators: separable-Gaussian stage 44 N0 + 48 sum N_o + input, whole image
    pipe 68 N0 + 72 sum N_o + input, for the 1080p workload (octave 0 = 3840 x 2160, 5 octaves)."""
    import bench
    px = bench.octave_pixels(1920, 1080, 5)
    assert px == [3840 * 2160, 1920 * 1080, 960 * 540, 480 * 270, 240 * 135]
    stage, pipe = bench.algorithmic_bytes(1920, 1080, 5)
    assert stage == 44 * px[0] + 48 * sum(px[1:]) + 1920 * 1080
    assert pipe == 68 * px[0] + 72 * sum(px[1:]) + 1920 * 1080
    assert abs(pipe - 0.764e9) < 1e6 and abs(stage - 499.2e6) < 1e5
    assert bench.BATCH * 20 >= 200                      # the driver's 20 steps time >= 200 frames per GPU
    assert bench.octave_pixels(75, 61, 3) == [150 * 122, 75 * 61, 38 * 31]


def test_bench_has_no_undefined_names():
    """bench.py's GPU-only legs cannot run here; at least every name they load must exist (a helper deleted by an edit
    would otherwise only surface on the GPU box)."""
    import ast
    import builtins
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    known = set(dir(builtins)) | {"__file__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            known.add(n.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            known.add(n.id)
        elif isinstance(n, ast.arg):
            known.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            known.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            known.add(n.name)
    loaded = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    assert loaded <= known, sorted(loaded - known)


def test_print_gauss_tables_prints_the_tables(capfd):
    """--print-gauss-tables / Config::setPrintGaussTables (gauss_filter.cu:24-120, 146-161): the five tables in the
    reference's layout; the numbers are those of psx_gauss_tables."""
    import ctypes as C
    from popsift_amd import capi
    L = capi.lib()
    L.psx_print_gauss_tables.argtypes = [C.POINTER(capi.Config), C.c_int]
    cfg = capi.default_config()
    assert L.psx_print_gauss_tables(C.byref(cfg), 10) == 0
    out = capfd.readouterr().out
    t = capi.gauss_tables(cfg)
    assert "Upscaling factor: 1.000000 (i.e. original image is scaled by a factor of 2.000000)" in out
    assert "    Initial sigma is 1.600000" in out and "Gauss tables for hardware interpolation" in out
    assert "absolute filters octave 0" in out and "level 0-filters for direct downscaling" in out
    rel = out.split("    relative sigma\n")[1].split("\n\n")[0].splitlines()
    assert len(rel) == cfg.levels + 3
    for lvl, line in enumerate(rel):
        head, vals = line.split(": ")
        assert head.split() == [str(lvl), str(2 * t["inc_span"][lvl] - 1), "%2.6f" % t["inc_sigma"][lvl]]
        want = ["%0.8f" % v for v in t["inc_filter"][lvl][:min(10, t["inc_span"][lvl])]]
        assert vals.split()[:len(want)] == want
        assert vals.rstrip().endswith("...") == (t["inc_span"][lvl] > 10)
    dd = out.split("level 0-filters for direct downscaling\n")[1].strip().splitlines()
    assert len(dd) == capi.MAX_OCTAVES and dd[0].split()[2] == "%2.6f:" % t["dd_sigma"][0]


@pytest.mark.parametrize("w0,h0,octaves,first", [(3840, 2160, 5, 0), (3840, 2160, 5, 1), (8192, 8192, 6, 0), (1280, 960, 5, 0),
                                                 (640, 480, 4, 0), (150, 122, 3, 0), (64, 64, 2, 0), (4097, 33, 3, 0), (9, 3000, 4, 0),
                                                 (2, 2, 1, 0), (1, 128, 2, 0), (667, 503, 4, 1)])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_pyramid_flow_plan_is_a_topological_order(w0, h0, octaves, first, order):
    """k_pyramid_flow's host-side plan (pyramid.hip psx_flow_plan), checked without a device: every (job, strip, chunk)
    exactly once, every consumer behind all of its producers in the ticket order (what the kernel's deadlock-freedom
    argument rests on), wait lists that cover every source row a workgroup loads, at most 64 counters per wait."""
    import ctypes as C
    from popsift_amd import capi
    L = capi.lib()
    L.psx_flow_selfcheck.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] + [C.c_int] * 3 + [C.POINTER(C.c_int)]
    t = capi.gauss_tables(capi.default_config())
    spans = (C.c_int * capi.GAUSS_LEVELS)(*[int(v) for v in t["inc_span"]])
    stats = (C.c_int * 4)()
    n = L.psx_flow_selfcheck(w0, h0, octaves, 3, spans, first, 1024, order, stats)
    assert n > 0, "psx_flow_selfcheck code %d" % n
    njobs, ncnt, grid, maxwait = list(stats)
    assert njobs == (octaves - first) * 5 and grid % 8 == 0 and 8 <= grid <= 1024 and 0 <= maxwait <= 64


def test_pyramid_flow_plan_refuses_large_radii():
    """A level whose radius exceeds 13 (the largest the flow kernel is instantiated for): the plan is refused and
    psx_build_pyramid keeps the launch-per-level schedule."""
    import ctypes as C
    from popsift_amd import capi
    L = capi.lib()
    L.psx_flow_selfcheck.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] + [C.c_int] * 3 + [C.POINTER(C.c_int)]
    spans = (C.c_int * capi.GAUSS_LEVELS)(*([6, 6, 8, 9, 11, 17] + [0] * (capi.GAUSS_LEVELS - 6)))
    assert L.psx_flow_selfcheck(640, 480, 3, 3, spans, 0, 1024, 0, None) == -2
    spans[5] = 14
    assert L.psx_flow_selfcheck(640, 480, 3, 3, spans, 0, 1024, 0, None) > 0


@pytest.mark.parametrize("w0,h0,octaves,levels", [(3840, 2160, 5, 3), (8192, 8192, 6, 3), (1280, 960, 5, 3), (640, 480, 4, 3),
                                                  (150, 122, 3, 3), (64, 64, 2, 3), (4097, 33, 3, 3), (9, 3000, 4, 3), (2, 2, 1, 3),
                                                  (1, 128, 2, 3), (667, 503, 4, 2), (667, 503, 4, 4), (1280, 960, 5, 5)])
@pytest.mark.parametrize("ty,nt", [(64, 512), (32, 512), (64, 1024)])
def test_tile_schedule_covers_every_level_once(w0, h0, octaves, levels, ty, nt):
    """k_blur_tile's host-side schedule (api.hip tile_schedule), checked without a device: every blur level of every octave
    from the first tiled one on is written by exactly one job, producers sit in earlier launches than their consumers, the
    decimated plane comes from the job that ends at level L-3, workgroup ranges of a launch are contiguous, LDS fits."""
    import ctypes as C
    from popsift_amd import capi
    L = capi.lib()
    L.psx_tile_selfcheck.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] + [C.c_int] * 2 + [C.c_longlong, C.POINTER(C.c_int)]
    t = capi.gauss_tables(capi.default_config(levels=levels))
    spans = (C.c_int * capi.GAUSS_LEVELS)(*[int(v) for v in t["inc_span"]])
    stats = (C.c_int * 4)()
    maxpx = 3 << 20
    n = L.psx_tile_selfcheck(w0, h0, octaves, levels, spans, ty, nt, maxpx, stats)
    assert n >= 0, "psx_tile_selfcheck code %d" % n
    first, launches, lds, grid = list(stats)
    small = [o for o in range(octaves) if ((w0 + (1 << o) - 1) >> o) * ((h0 + (1 << o) - 1) >> o) <= maxpx]
    if n == 0:
        assert not small or max(int(v) for v in spans[:levels + 3]) - 1 > 13
    else:
        assert first == small[0] and lds <= 160 * 1024
        if levels == 3:
            # default configuration: levels 1..3 and 4..5 of every tiled octave, octave o's second job beside octave o+1's first
            assert n == 2 * (octaves - first) and launches == octaves - first + 1


def test_tile_schedule_refuses_large_radii():
    import ctypes as C
    from popsift_amd import capi
    L = capi.lib()
    L.psx_tile_selfcheck.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] + [C.c_int] * 2 + [C.c_longlong, C.POINTER(C.c_int)]
    spans = (C.c_int * capi.GAUSS_LEVELS)(*([6, 6, 8, 9, 11, 17] + [0] * (capi.GAUSS_LEVELS - 6)))
    assert L.psx_tile_selfcheck(640, 480, 3, 3, spans, 64, 512, 3 << 20, None) == 0
    spans[5] = 14
    assert L.psx_tile_selfcheck(640, 480, 3, 3, spans, 64, 512, 3 << 20, None) == 6
