"""ctypes binding of oracle/_ref/libpopsift_ref.so: the REFERENCE's own sources running on the CPU
through the CUDA emulation in oracle/ref_shim.  TEST INFRASTRUCTURE ONLY.

Only available where /root/reference exists at build time (`make -C oracle ref`); the built .so
travels with the repo snapshot, the sources never do.
"""
import ctypes as C
import os

import numpy as np

from . import pyoracle as po

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSIFT_REF_LIB: a variant build (make -C oracle ref_nofma / ref_model)
SO = os.environ.get("OSIFT_REF_LIB") or os.path.join(_HERE, "_ref", "libpopsift_ref.so")
_LIB = None


def available():
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(SO)
        for name in ("ref_run", "ref_run_api"):
            f = getattr(L, name)
            f.argtypes = [C.POINTER(po.Config), C.c_void_p, C.c_int, C.c_int, C.c_int]
            f.restype = C.c_void_p
        L.ref_gauss_tables.argtypes = [C.POINTER(po.Config), C.POINTER(po.Tables)]
        L.ref_peak_threshold.argtypes = [C.POINTER(po.Config)]
        L.ref_peak_threshold.restype = C.c_float
        L.ref_free.argtypes = [C.c_void_p]
        for name in ("ref_num_octaves", "ref_num_levels", "ref_ext_total", "ref_ori_total"):
            f = getattr(L, name); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        for name in ("ref_octave_width", "ref_octave_height", "ref_iext_count"):
            f = getattr(L, name); f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
        for name in ("ref_gauss_plane", "ref_dog_plane"):
            f = getattr(L, name); f.argtypes = [C.c_void_p, C.c_int, C.c_int]; f.restype = C.c_void_p
        L.ref_get_iext.argtypes = [C.c_void_p, C.c_int]; L.ref_get_iext.restype = C.c_void_p
        for name in ("ref_features", "ref_descriptors"):
            f = getattr(L, name); f.argtypes = [C.c_void_p]; f.restype = C.c_void_p
        _LIB = L
    return _LIB


def gauss_tables(cfg):
    t = po.Tables()
    if lib().ref_gauss_tables(C.byref(cfg), C.byref(t)) != 0:
        raise RuntimeError("reference init_filter threw")
    return {
        "inc_filter": np.array(t.inc_filter, dtype=np.float32).reshape(po.GAUSS_LEVELS, po.GAUSS_ALIGN),
        "inc_sigma": np.array(t.inc_sigma, dtype=np.float32),
        "inc_span": np.array(t.inc_span, dtype=np.int32),
        "dd_filter": np.array(t.dd_filter, dtype=np.float32).reshape(po.MAX_OCTAVES, po.GAUSS_ALIGN),
        "dd_sigma": np.array(t.dd_sigma, dtype=np.float32),
        "dd_span": np.array(t.dd_span, dtype=np.int32),
        "abs0_filter": np.array(t.abs0_filter, dtype=np.float32).reshape(po.GAUSS_LEVELS, po.GAUSS_ALIGN),
        "abs0_sigma": np.array(t.abs0_sigma, dtype=np.float32),
        "abs0_span": np.array(t.abs0_span, dtype=np.int32),
        "absN_filter": np.array(t.absN_filter, dtype=np.float32).reshape(po.GAUSS_LEVELS, po.GAUSS_ALIGN),
        "absN_sigma": np.array(t.absN_sigma, dtype=np.float32),
        "absN_span": np.array(t.absN_span, dtype=np.int32),
        "inc_ifilter": np.array(t.inc_ifilter, dtype=np.float32).reshape(po.GAUSS_LEVELS, po.GAUSS_ALIGN),
        "inc_ispan": np.array(t.inc_ispan, dtype=np.int32),
    }


class Result:
    def __init__(self, h, full=True):
        if not h:
            raise RuntimeError("reference run failed")
        self._h = h
        L = lib()
        self.num_octaves = L.ref_num_octaves(h)
        self.num_levels = L.ref_num_levels(h)
        self.dims = [(L.ref_octave_width(h, o), L.ref_octave_height(h, o)) for o in range(self.num_octaves)]
        self.ext_total = L.ref_ext_total(h)
        self.ori_total = L.ref_ori_total(h)

    def gauss(self, o, l):
        w, h = self.dims[o]
        return po._arr(lib().ref_gauss_plane(self._h, o, l), np.float32, w * h).reshape(h, w)

    def dog(self, o, l):
        w, h = self.dims[o]
        return po._arr(lib().ref_dog_plane(self._h, o, l), np.float32, w * h).reshape(h, w)

    def iext(self, o):
        return po._arr(lib().ref_get_iext(self._h, o), po.IEXT_DTYPE, lib().ref_iext_count(self._h, o))

    def features(self):
        return po._arr(lib().ref_features(self._h), po.FEAT_DTYPE, self.ext_total)

    def descriptors(self):
        return po._arr(lib().ref_descriptors(self._h), np.float32, self.ori_total * 128).reshape(-1, 128)

    def close(self):
        if self._h:
            lib().ref_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run(cfg, img, api=False):
    img, w, h, is_float = po._img_args(img)
    f = lib().ref_run_api if api else lib().ref_run
    return Result(f(C.byref(cfg), img.ctypes.data_as(C.c_void_p), w, h, is_float))


def match(left, right):
    """The reference's FeaturesDev::match (its own features.cu on the CUDA emulation); results parsed
    from what it prints.  Returns (n,3) int32 {best, second, accept} and (n,2) float32 printed distances."""
    left = np.ascontiguousarray(left, dtype=np.float32).reshape(-1, 128)
    right = np.ascontiguousarray(right, dtype=np.float32).reshape(-1, 128)
    L = lib()
    L.ref_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    mm = np.full((len(left), 3), -1, np.int32)
    dd = np.zeros((len(left), 2), np.float32)
    n = L.ref_match(left.ctypes.data_as(C.c_void_p), len(left), right.ctypes.data_as(C.c_void_p), len(right),
                    mm.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p))
    if n != len(left):
        raise RuntimeError("ref_match parsed %d of %d lines" % (n, len(left)))
    return mm, dd
