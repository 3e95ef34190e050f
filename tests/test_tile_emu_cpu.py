"""The multi-level tile kernel (k_blur_tile) without a GPU: its phase functions, compiled for the host, against the oracle.

popsift_amd/csrc/hip/blur_tile_core.h holds the phases of k_blur_tile (loader, H pass, V pass, edge re-clamping) as
functions of the thread index; tests/cpp/tile_emu.cpp compiles the same header for the CPU and runs every tile, phase
and thread serially, with LDS and the destination planes pre-set to NaN.  The planes must equal the oracle's bit for
bit: that checks the plan (halos, regions, alignment), the lane -> cell mappings, the clamping and the store predicates
-- everything but the device-only macros.  The GPU test (tests/test_gpu_parity.py::test_tile_kernel_*) checks the rest.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from popsift_amd.synth import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ (ext_vector_type) in this image")
    so = str(tmp_path_factory.mktemp("tile_emu") / "libtile_emu.so")
    subprocess.check_call([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-DPSX_TILE_EMU", "-shared", "-fPIC",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "popsift_amd", "csrc", "hip"),
                           os.path.join(ROOT, "tests", "cpp", "tile_emu.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.tile_emu_run.restype = C.c_int
    lib.tile_emu_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return lib


def pitched(plane, fill=np.nan):
    h, w = plane.shape
    pitch = (w + 63) & ~63
    buf = np.full((h, pitch), fill, np.float32)
    buf[:, :w] = plane
    return buf, pitch


def run_job(emu, ref, tabs, o, l0, nlev, tx, ty, nt, with_half):
    """levels l0 .. l0+nlev-1 of octave o from the oracle's plane l0-1; returns the number of mismatching pixels"""
    w, h = ref.dims[o]
    src, pitch = pitched(ref.gauss(o, l0 - 1))          # NaN in the pad columns: nobody may read them
    spans = np.ascontiguousarray(tabs["inc_span"][l0:l0 + nlev], np.int32)
    taps = np.ascontiguousarray(tabs["inc_filter"][l0:l0 + nlev], np.float32)
    dst = np.full((nlev, h, pitch), np.nan, np.float32)
    half = None
    half_lev, half_pitch = -1, 0
    L = ref.num_levels
    if with_half and o + 1 < ref.num_octaves and l0 <= L - 3 < l0 + nlev:
        w2, h2 = ref.dims[o + 1]
        half_pitch = (w2 + 63) & ~63
        half = np.full((h2, half_pitch), np.nan, np.float32)
        half_lev = L - 3 - l0
    info = np.zeros(8, np.int32)
    rc = emu.tile_emu_run(src.ctypes.data, w, h, pitch, nlev, spans.ctypes.data, taps.ctypes.data, tx, ty, nt,
                          dst.ctypes.data, half.ctypes.data if half is not None else None, half_lev, half_pitch,
                          info.ctypes.data)
    if rc != 0:
        return None, info
    bad = 0
    for k in range(nlev):
        want = ref.gauss(o, l0 + k)
        got = dst[k, :, :w]
        bad += int(np.count_nonzero(want.view(np.uint32) != got.view(np.uint32)))
        # the pad columns stay untouched
        assert np.all(np.isnan(dst[k, :, w:]))
    if half is not None:
        w2, h2 = ref.dims[o + 1]
        want = ref.gauss(o + 1, 0)
        bad += int(np.count_nonzero(want.view(np.uint32) != half[:, :w2].view(np.uint32)))
    return bad, info


CASES = [
    # (w, h, octaves, sift mode, tile columns, tile rows, threads)
    (200, 150, 3, po.MODE_POPSIFT, 64, 64, 512),      # x2: octave 0 is 400 x 300 -> tiles 7 x 5, octaves of 200x150, 100x75
    (163, 122, 3, po.MODE_VLFEAT, 64, 32, 512),       # odd sizes: last tiles partly outside the plane, one-pixel columns
    (96, 64, 2, po.MODE_OPENCV, 64, 64, 1024),        # OpenCV spans (radius 6 runs on the radius-7 body)
    (33, 21, 2, po.MODE_POPSIFT, 64, 32, 1024),       # planes smaller than one tile's halo
    (163, 122, 3, po.MODE_POPSIFT, 32, 32, 1024),     # the 32 x 32 tiles of the octaves that cannot fill the chip
    (120, 90, 2, po.MODE_OPENCV, 32, 32, 512),
]


@pytest.mark.parametrize("w,h,octaves,mode,tx,ty,nt", CASES)
def test_tile_phases_reproduce_the_oracle_planes(emu, w, h, octaves, mode, tx, ty, nt):
    cfg = po.default_config(octaves=octaves, sift_mode=mode)
    if mode == po.MODE_OPENCV:
        cfg.gauss_mode = po.GAUSS_OPENCV_COMPUTE
    img = synth(w, h, 7)
    ref = po.run_pyramid(cfg, img)
    tabs = po.gauss_tables(cfg)
    L = ref.num_levels
    D = L - 3
    ran = 0
    for o in range(ref.num_octaves):
        # the two jobs of the default schedule: levels 1..L-3 (+ decimation), levels L-2..L-1
        for l0, n, half in ((1, D, True), (D + 1, L - 1 - D, False)):
            bad, info = run_job(emu, ref, tabs, o, l0, n, tx, ty, nt, half)
            assert bad is not None, "plan rejected a default-config job: %s" % info
            assert bad == 0, "octave %d levels %d..%d: %d pixels differ (plan %s)" % (o, l0, l0 + n - 1, bad, info)
            ran += 1
    assert ran == 2 * ref.num_octaves


def test_all_levels_in_one_job_and_single_levels(emu):
    """other splits of the level range: every level alone, and all five at once (rejected or exact, never wrong)"""
    cfg = po.default_config(octaves=2)
    img = synth(120, 90, 3)
    ref = po.run_pyramid(cfg, img)
    tabs = po.gauss_tables(cfg)
    L = ref.num_levels
    for l in range(1, L):
        bad, info = run_job(emu, ref, tabs, 1, l, 1, 64, 64, 512, True)
        assert bad == 0, (l, info)
    bad, info = run_job(emu, ref, tabs, 1, 1, L - 1, 64, 32, 1024, True)
    assert bad is None or bad == 0                 # five levels need 43 halo columns: more than Q's 96 columns hold
    bad, info = run_job(emu, ref, tabs, 0, 2, 3, 64, 64, 1024, True)
    assert bad is None or bad == 0
