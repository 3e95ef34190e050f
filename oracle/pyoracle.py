"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg and the parity_checked step (checker, outside the timed regions) of bench.py -- never from popsift_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_OCTAVES = 20
GAUSS_ALIGN = 32
GAUSS_LEVELS = 12
ORI_MAX = 4

GAUSS_VLFEAT_COMPUTE, GAUSS_VLFEAT_RELATIVE, GAUSS_VLFEAT_RELATIVE_ALL, GAUSS_OPENCV_COMPUTE, \
    GAUSS_FIXED9, GAUSS_FIXED15 = range(6)
MODE_POPSIFT, MODE_OPENCV, MODE_VLFEAT = 0, 1, 2
NORM_ROOTSIFT, NORM_CLASSIC = 0, 1
SCALE_DIRECT, SCALE_DEFAULT = 0, 1
DESC_LOOP, DESC_ILOOP, DESC_GRID, DESC_IGRID, DESC_NOTILE = range(5)
FILTER_RANDOM, FILTER_LARGEST_FIRST, FILTER_SMALLEST_FIRST = 0, 1, 2


class Config(C.Structure):
    _fields_ = [
        ("octaves", C.c_int), ("levels", C.c_int), ("sigma", C.c_float),
        ("edge_limit", C.c_float), ("threshold", C.c_float), ("upscale_factor", C.c_float),
        ("gauss_mode", C.c_int), ("sift_mode", C.c_int), ("norm_mode", C.c_int),
        ("norm_multi", C.c_int), ("max_extrema", C.c_int), ("assume_initial_blur", C.c_int),
        ("initial_blur", C.c_float), ("filter_max_extrema", C.c_int),
        ("filter_grid_size", C.c_int), ("grid_filter_mode", C.c_int), ("literal_tex", C.c_int),
        ("scaling_mode", C.c_int), ("desc_mode", C.c_int),
    ]


class Tables(C.Structure):
    _fields_ = [
        ("inc_filter", C.c_float * (GAUSS_LEVELS * GAUSS_ALIGN)),
        ("inc_sigma", C.c_float * GAUSS_LEVELS),
        ("inc_span", C.c_int * GAUSS_LEVELS),
        ("dd_filter", C.c_float * (MAX_OCTAVES * GAUSS_ALIGN)),
        ("dd_sigma", C.c_float * MAX_OCTAVES),
        ("dd_span", C.c_int * MAX_OCTAVES),
        ("abs0_filter", C.c_float * (GAUSS_LEVELS * GAUSS_ALIGN)),
        ("abs0_sigma", C.c_float * GAUSS_LEVELS),
        ("abs0_span", C.c_int * GAUSS_LEVELS),
        ("absN_filter", C.c_float * (GAUSS_LEVELS * GAUSS_ALIGN)),
        ("absN_sigma", C.c_float * GAUSS_LEVELS),
        ("absN_span", C.c_int * GAUSS_LEVELS),
        ("inc_ifilter", C.c_float * (GAUSS_LEVELS * GAUSS_ALIGN)),
        ("inc_ispan", C.c_int * GAUSS_LEVELS),
    ]


IEXT_DTYPE = np.dtype([("xpos", "<f4"), ("ypos", "<f4"), ("lpos", "<i4"), ("sigma", "<f4"),
                       ("cell", "<i4"), ("ignore", "<i4")])
EXT_DTYPE = np.dtype([("xpos", "<f4"), ("ypos", "<f4"), ("lpos", "<i4"), ("sigma", "<f4"),
                      ("octave", "<i4"), ("num_ori", "<i4"), ("idx_ori", "<i4"),
                      ("orientation", "<f4", (ORI_MAX,))])
FEAT_DTYPE = np.dtype([("debug_octave", "<i4"), ("xpos", "<f4"), ("ypos", "<f4"), ("sigma", "<f4"),
                       ("num_ori", "<i4"), ("orientation", "<f4", (ORI_MAX,)),
                       ("desc_idx", "<i4", (ORI_MAX,))])


def build(force=False):
    """Compile liboracle.so (gcc, a second or two)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("sift_oracle.c", "sift_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("OSIFT_LIB") or os.path.join(_HERE, "liboracle.so")     # OSIFT_LIB: a variant build (make plain)
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.osift_config_default.argtypes = [C.POINTER(Config)]
        L.osift_peak_threshold.argtypes = [C.POINTER(Config)]
        L.osift_peak_threshold.restype = C.c_float
        L.osift_gauss_tables.argtypes = [C.POINTER(Config), C.POINTER(Tables)]
        L.osift_gauss_tables.restype = C.c_int
        for name in ("osift_run", "osift_run_pyramid"):
            f = getattr(L, name)
            f.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int, C.c_int, C.c_int]
            f.restype = C.c_void_p
        L.osift_free.argtypes = [C.c_void_p]
        for name in ("osift_num_octaves", "osift_num_levels", "osift_ext_total", "osift_ori_total"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_int
        for name in ("osift_octave_width", "osift_octave_height", "osift_iext_count"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p, C.c_int]
            f.restype = C.c_int
        for name in ("osift_gauss_plane", "osift_dog_plane"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p, C.c_int, C.c_int]
            f.restype = C.c_void_p
        L.osift_get_iext.argtypes = [C.c_void_p, C.c_int]
        L.osift_get_iext.restype = C.c_void_p
        for name in ("osift_extrema", "osift_features", "osift_descriptors", "osift_feat_to_ext"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_void_p
        L.osift_describe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.osift_describe.restype = C.c_int
        L.osift_set_threads.argtypes = [C.c_int]
        L.osift_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.osift_match.restype = None
        _LIB = L
    return _LIB


def time_fast_build(cfg, imgs, threads):
    """bench.py's cpu_baseline only: the same source built -O3 -march=native -ffp-contract=fast on THIS host
    (oracle/_fast/liboracle_fast.so, `make fast`; not the checker -- the strict build above stays the checker) and
    timed over imgs.  Returns seconds, or None when the build is not possible here."""
    import time
    try:
        subprocess.check_call(["make", "-C", _HERE, "fast"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        F = C.CDLL(os.path.join(_HERE, "_fast", "liboracle_fast.so"))
    except Exception:
        return None
    F.osift_run.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int, C.c_int, C.c_int]
    F.osift_run.restype = C.c_void_p
    F.osift_free.argtypes = [C.c_void_p]
    F.osift_set_threads.argtypes = [C.c_int]
    F.osift_set_threads(threads)
    prepared = [_img_args(i) for i in imgs]
    im, w, h, fl = prepared[0]
    F.osift_free(F.osift_run(C.byref(cfg), im.ctypes.data_as(C.c_void_p), w, h, fl))       # warm
    t0 = time.perf_counter()
    for im, w, h, fl in prepared:
        F.osift_free(F.osift_run(C.byref(cfg), im.ctypes.data_as(C.c_void_p), w, h, fl))
    return time.perf_counter() - t0


def default_config(**kw):
    c = Config()
    lib().osift_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def gauss_tables(cfg):
    t = Tables()
    rc = lib().osift_gauss_tables(C.byref(cfg), C.byref(t))
    if rc != 0:
        raise RuntimeError("osift_gauss_tables failed: %d" % rc)
    return {
        "inc_filter": np.array(t.inc_filter, dtype=np.float32).reshape(GAUSS_LEVELS, GAUSS_ALIGN),
        "inc_sigma": np.array(t.inc_sigma, dtype=np.float32),
        "inc_span": np.array(t.inc_span, dtype=np.int32),
        "dd_filter": np.array(t.dd_filter, dtype=np.float32).reshape(MAX_OCTAVES, GAUSS_ALIGN),
        "dd_sigma": np.array(t.dd_sigma, dtype=np.float32),
        "dd_span": np.array(t.dd_span, dtype=np.int32),
        "abs0_filter": np.array(t.abs0_filter, dtype=np.float32).reshape(GAUSS_LEVELS, GAUSS_ALIGN),
        "abs0_sigma": np.array(t.abs0_sigma, dtype=np.float32),
        "abs0_span": np.array(t.abs0_span, dtype=np.int32),
        "absN_filter": np.array(t.absN_filter, dtype=np.float32).reshape(GAUSS_LEVELS, GAUSS_ALIGN),
        "absN_sigma": np.array(t.absN_sigma, dtype=np.float32),
        "absN_span": np.array(t.absN_span, dtype=np.int32),
        "inc_ifilter": np.array(t.inc_ifilter, dtype=np.float32).reshape(GAUSS_LEVELS, GAUSS_ALIGN),
        "inc_ispan": np.array(t.inc_ispan, dtype=np.int32),
    }


def _arr(ptr, dtype, count):
    if count == 0 or not ptr:
        return np.zeros((0,), dtype=dtype)
    nbytes = np.dtype(dtype).itemsize * count
    buf = (C.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class Result:
    """Owning view of an osift_result."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle run failed")
        self._h = handle
        L = lib()
        self.num_octaves = L.osift_num_octaves(handle)
        self.num_levels = L.osift_num_levels(handle)
        self.dims = [(L.osift_octave_width(handle, o), L.osift_octave_height(handle, o))
                     for o in range(self.num_octaves)]

    def gauss(self, o, l):
        w, h = self.dims[o]
        return _arr(lib().osift_gauss_plane(self._h, o, l), np.float32, w * h).reshape(h, w)

    def dog(self, o, l):
        w, h = self.dims[o]
        return _arr(lib().osift_dog_plane(self._h, o, l), np.float32, w * h).reshape(h, w)

    def iext(self, o):
        n = lib().osift_iext_count(self._h, o)
        return _arr(lib().osift_get_iext(self._h, o), IEXT_DTYPE, n)

    @property
    def ext_total(self):
        return lib().osift_ext_total(self._h)

    @property
    def ori_total(self):
        return lib().osift_ori_total(self._h)

    def extrema(self):
        return _arr(lib().osift_extrema(self._h), EXT_DTYPE, self.ext_total)

    def features(self):
        return _arr(lib().osift_features(self._h), FEAT_DTYPE, self.ext_total)

    def descriptors(self):
        return _arr(lib().osift_descriptors(self._h), np.float32, self.ori_total * 128).reshape(-1, 128)

    def feat_to_ext(self):
        return _arr(lib().osift_feat_to_ext(self._h), np.int32, self.ori_total)

    def describe(self, extrema, n_desc):
        """Descriptor stage alone on this result's pyramid for caller-supplied oriented extrema (EXT_DTYPE records,
        e.g. a device run's psx_dump_extrema): (n_desc, 128) array indexed by idx_ori + k."""
        ext = np.ascontiguousarray(extrema)
        assert ext.dtype.itemsize == EXT_DTYPE.itemsize
        out = np.zeros((max(n_desc, 1), 128), np.float32)
        lib().osift_describe(self._h, ext.ctypes.data, len(ext), n_desc, out.ctypes.data)
        return out[:n_desc]

    def close(self):
        if self._h:
            lib().osift_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _img_args(img):
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint8:
        is_float = 0
    elif img.dtype == np.float32:
        is_float = 1
    else:
        raise TypeError("image must be uint8 or float32")
    h, w = img.shape
    return img, w, h, is_float


def default_threads():
    """OpenMP threads for the oracle: usable cores (affinity and cgroup quota), at most 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def run(cfg, img, threads=0):
    img, w, h, is_float = _img_args(img)
    lib().osift_set_threads(threads if threads > 0 else default_threads())
    return Result(lib().osift_run(C.byref(cfg), img.ctypes.data_as(C.c_void_p), w, h, is_float))


def run_pyramid(cfg, img, threads=0):
    img, w, h, is_float = _img_args(img)
    lib().osift_set_threads(threads if threads > 0 else default_threads())
    return Result(lib().osift_run_pyramid(C.byref(cfg), img.ctypes.data_as(C.c_void_p), w, h, is_float))


def match(left, right, threads=0):
    """osift_match: (n,3) int32 {best, second, accept} and (n,2) float32 squared distances."""
    left = np.ascontiguousarray(left, dtype=np.float32).reshape(-1, 128)
    right = np.ascontiguousarray(right, dtype=np.float32).reshape(-1, 128)
    lib().osift_set_threads(threads if threads > 0 else default_threads())
    mm = np.zeros((len(left), 3), np.int32)
    dd = np.zeros((len(left), 2), np.float32)
    lib().osift_match(left.ctypes.data_as(C.c_void_p), len(left), right.ctypes.data_as(C.c_void_p), len(right),
                      mm.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p))
    return mm, dd
