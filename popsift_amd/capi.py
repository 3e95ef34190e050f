"""ctypes binding of the C-ABI (include/popsift_hip.h) -> popsift_amd/lib/libpopsift_hip.so.

This is plumbing for tests and bench.py; the product host side is the C++14 library
(popsift_amd/csrc/host).  The binding fails loudly when the HIP library is missing: there is
no CPU fallback anywhere in the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POPSIFT_HIP_LIB") or os.path.join(_HERE, "lib", "libpopsift_hip.so")

MAX_OCTAVES = 20
GAUSS_ALIGN = 32
GAUSS_LEVELS = 12
ORI_MAX = 4

GAUSS_VLFEAT_COMPUTE, GAUSS_VLFEAT_RELATIVE, GAUSS_VLFEAT_RELATIVE_ALL, GAUSS_OPENCV_COMPUTE, \
    GAUSS_FIXED9, GAUSS_FIXED15 = range(6)
MODE_POPSIFT, MODE_OPENCV, MODE_VLFEAT = 0, 1, 2
SCALE_DIRECT, SCALE_DEFAULT = 0, 1
DESC_LOOP, DESC_ILOOP, DESC_GRID, DESC_IGRID, DESC_NOTILE = range(5)
NORM_ROOTSIFT, NORM_CLASSIC = 0, 1
FILTER_RANDOM, FILTER_LARGEST_FIRST, FILTER_SMALLEST_FIRST = 0, 1, 2
PLANE_GAUSS, PLANE_DOG = 0, 1


class Config(C.Structure):
    """POD image of popsift::Config (psx_config)."""
    _fields_ = [
        ("octaves", C.c_int), ("levels", C.c_int), ("sigma", C.c_float), ("edge_limit", C.c_float),
        ("threshold", C.c_float), ("upscale_factor", C.c_float), ("gauss_mode", C.c_int),
        ("sift_mode", C.c_int), ("scaling_mode", C.c_int), ("desc_mode", C.c_int),
        ("norm_mode", C.c_int), ("norm_multi", C.c_int), ("max_extrema", C.c_int),
        ("assume_initial_blur", C.c_int), ("initial_blur", C.c_float),
        ("filter_max_extrema", C.c_int), ("filter_grid_size", C.c_int), ("grid_filter_mode", C.c_int),
    ]


FEATURE_DTYPE = np.dtype([("debug_octave", "<i4"), ("xpos", "<f4"), ("ypos", "<f4"), ("sigma", "<f4"),
                          ("num_ori", "<i4"), ("orientation", "<f4", (ORI_MAX,)),
                          ("desc_idx", "<i4", (ORI_MAX,))])
# popsift::Feature as stored in a FeaturesDev (psx_feature_dev, 72 bytes; desc[] are device addresses)
FEATURE_DEV_DTYPE = np.dtype([("debug_octave", "<i4"), ("xpos", "<f4"), ("ypos", "<f4"), ("sigma", "<f4"),
                              ("num_ori", "<i4"), ("orientation", "<f4", (ORI_MAX,)), ("pad", "<i4"),
                              ("desc", "<u8", (ORI_MAX,))])
IEXT_DTYPE = np.dtype([("xpos", "<f4"), ("ypos", "<f4"), ("lpos", "<i4"), ("sigma", "<f4"),
                       ("cell", "<i4"), ("ignore", "<i4")])
EXT_DTYPE = np.dtype([("xpos", "<f4"), ("ypos", "<f4"), ("lpos", "<i4"), ("sigma", "<f4"),
                      ("octave", "<i4"), ("num_ori", "<i4"), ("idx_ori", "<i4"),
                      ("orientation", "<f4", (ORI_MAX,))])

# every symbol include/popsift_hip.h declares
SYMBOLS = [
    "psx_version", "psx_config_default", "psx_peak_threshold", "psx_gauss_tables", "psx_create",
    "psx_destroy", "psx_last_error", "psx_resize", "psx_num_octaves", "psx_num_levels",
    "psx_octave_dims", "psx_upload_u8", "psx_upload_f32", "psx_set_input_dev", "psx_build_pyramid",
    "psx_find_extrema", "psx_orientation", "psx_descriptors", "psx_extract", "psx_sync", "psx_counts",
    "psx_download", "psx_attach_export", "psx_device_results", "psx_dump_plane", "psx_dump_iext", "psx_dump_extrema",
    "psx_set_wait_mode", "psx_enable_timers", "psx_stage_times", "psx_time_blur", "psx_stream",
    "psx_host_alloc", "psx_host_alloc_near", "psx_host_free", "psx_dev_alloc", "psx_dev_free", "psx_dev_read", "psx_dev_write", "psx_clone_results", "psx_match", "psx_match_release", "psx_device_count", "psx_device_info", "psx_device_pci",
    "psx_enable_blur_probe", "psx_blur_probe_times", "psx_copy_bench", "psx_upload_pinned", "psx_attach_export_mapped",
    "psx_print_gauss_tables", "psx_flow_trace", "psx_debug_cross_stream", "psx_probe_extra_times",
]

_LIB = None


class PopSiftError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PopSiftError(
                "%s is missing: build it with `python -m popsift_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.psx_version.restype = C.c_char_p
        L.psx_config_default.argtypes = [C.POINTER(Config)]
        L.psx_peak_threshold.argtypes = [C.POINTER(Config)]
        L.psx_peak_threshold.restype = C.c_float
        L.psx_gauss_tables.argtypes = [C.POINTER(Config), fp, ip, fp, fp, ip, fp]
        L.psx_create.argtypes = [C.c_int, C.POINTER(Config), C.POINTER(vp)]
        L.psx_destroy.argtypes = [vp]
        L.psx_last_error.argtypes = [vp]
        L.psx_last_error.restype = C.c_char_p
        L.psx_resize.argtypes = [vp, C.c_int, C.c_int]
        L.psx_num_octaves.argtypes = [vp]
        L.psx_num_levels.argtypes = [vp]
        L.psx_octave_dims.argtypes = [vp, C.c_int, ip, ip]
        L.psx_upload_u8.argtypes = [vp, vp, C.c_int, C.c_int]
        L.psx_upload_f32.argtypes = [vp, vp, C.c_int, C.c_int]
        L.psx_set_input_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
        for n in ("psx_build_pyramid", "psx_find_extrema", "psx_orientation", "psx_descriptors",
                  "psx_extract", "psx_sync"):
            getattr(L, n).argtypes = [vp]
        L.psx_counts.argtypes = [vp, ip, ip]
        L.psx_download.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.psx_attach_export.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.psx_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.psx_dump_plane.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.psx_dump_iext.argtypes = [vp, C.c_int, vp, C.c_int, ip]
        L.psx_dump_extrema.argtypes = [vp, vp, C.c_int, ip]
        L.psx_enable_timers.argtypes = [vp, C.c_int]
        L.psx_stage_times.argtypes = [vp, fp]
        L.psx_time_blur.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, C.POINTER(C.c_double)]
        L.psx_stream.argtypes = [vp]
        L.psx_stream.restype = vp
        L.psx_enable_blur_probe.argtypes = [vp, C.c_int]
        L.psx_blur_probe_times.argtypes = [vp, fp, C.c_int, ip, C.POINTER(C.c_double)]
        L.psx_copy_bench.argtypes = [C.c_int, C.c_size_t, C.c_int, fp, C.POINTER(C.c_double)]
        L.psx_device_pci.argtypes = [C.c_int, C.c_char_p, C.c_int]
        _LIB = L
    return _LIB


def default_config(**kw):
    c = Config()
    lib().psx_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def match(left, right, device=0):
    """psx_match on host arrays: left (n,128) / right (m,128) float32 go to the device through
    psx_dev_alloc / psx_dev_write.  Returns (match (n,3) int32 = best, second, accept; dist (n,2) float32
    squared distances)."""
    L = lib()
    L.psx_match.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.psx_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.psx_dev_free.argtypes = [C.c_int, C.c_void_p]
    L.psx_dev_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    left = np.ascontiguousarray(left, dtype=np.float32).reshape(-1, 128)
    right = np.ascontiguousarray(right, dtype=np.float32).reshape(-1, 128)
    bufs = []
    try:
        ptrs = []
        for arr in (left, right):
            p = C.c_void_p()
            if len(arr):
                if L.psx_dev_alloc(device, arr.nbytes, C.byref(p)) != 0:
                    raise PopSiftError("psx_dev_alloc failed")
                bufs.append(p)
                if L.psx_dev_write(device, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes) != 0:
                    raise PopSiftError("psx_dev_write failed")
            ptrs.append(p)
        mm = np.zeros((len(left), 3), np.int32)
        dd = np.zeros((len(left), 2), np.float32)
        rc = L.psx_match(device, ptrs[0], len(left), ptrs[1], len(right),
                         mm.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise PopSiftError("psx_match failed (%d)" % rc)
        return mm, dd
    finally:
        for p in bufs:
            L.psx_dev_free(device, p)


class DeviceDescriptors:
    """Descriptor sets resident on the device (what FeaturesDev holds, popsift.cpp:346-383): upload once, match many times."""

    def __init__(self, arr, device=0):
        L = lib()
        L.psx_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.psx_dev_free.argtypes = [C.c_int, C.c_void_p]
        L.psx_dev_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 128)
        self.n, self.device, self.ptr = len(arr), device, C.c_void_p()
        if self.n:
            if L.psx_dev_alloc(device, arr.nbytes, C.byref(self.ptr)) != 0:
                raise PopSiftError("psx_dev_alloc failed")
            if L.psx_dev_write(device, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes) != 0:
                raise PopSiftError("psx_dev_write failed")

    def match(self, right):
        """psx_match(self as left, right: DeviceDescriptors): (n,3) int32 and (n,2) float32 host arrays"""
        L = lib()
        L.psx_match.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        mm = np.zeros((self.n, 3), np.int32)
        dd = np.zeros((self.n, 2), np.float32)
        rc = L.psx_match(self.device, self.ptr, self.n, right.ptr, right.n, mm.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise PopSiftError("psx_match failed (%d)" % rc)
        return mm, dd

    def close(self):
        if self.ptr:
            lib().psx_dev_free(self.device, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gauss_tables(cfg):
    inc_f = (C.c_float * (GAUSS_LEVELS * GAUSS_ALIGN))()
    inc_sp = (C.c_int * GAUSS_LEVELS)()
    inc_sg = (C.c_float * GAUSS_LEVELS)()
    dd_f = (C.c_float * (MAX_OCTAVES * GAUSS_ALIGN))()
    dd_sp = (C.c_int * MAX_OCTAVES)()
    dd_sg = (C.c_float * MAX_OCTAVES)()
    rc = lib().psx_gauss_tables(C.byref(cfg), inc_f, inc_sp, inc_sg, dd_f, dd_sp, dd_sg)
    if rc != 0:
        raise PopSiftError("psx_gauss_tables failed (%d)" % rc)
    return {
        "inc_filter": np.array(inc_f, dtype=np.float32).reshape(GAUSS_LEVELS, GAUSS_ALIGN),
        "inc_sigma": np.array(inc_sg, dtype=np.float32),
        "inc_span": np.array(inc_sp, dtype=np.int32),
        "dd_filter": np.array(dd_f, dtype=np.float32).reshape(MAX_OCTAVES, GAUSS_ALIGN),
        "dd_sigma": np.array(dd_sg, dtype=np.float32),
        "dd_span": np.array(dd_sp, dtype=np.int32),
    }


class Context:
    """One extraction context (pyramid + buffers + HIP stream) on one device."""

    def __init__(self, cfg=None, device=0):
        self._h = C.c_void_p()
        self.cfg = cfg if cfg is not None else default_config()
        rc = lib().psx_create(device, C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            msg = lib().psx_last_error(None)
            raise PopSiftError("psx_create failed (%d): %s" % (rc, msg.decode() if msg else ""))
        self._keep = None

    def _chk(self, rc):
        if rc != 0:
            msg = lib().psx_last_error(self._h)
            raise PopSiftError("C-ABI call failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def close(self):
        if self._h:
            lib().psx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- input -------------------------------------------------------------------------------
    def upload(self, img):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        if img.dtype == np.uint8:
            self._chk(lib().psx_upload_u8(self._h, img.ctypes.data_as(C.c_void_p), w, h))
        elif img.dtype == np.float32:
            self._chk(lib().psx_upload_f32(self._h, img.ctypes.data_as(C.c_void_p), w, h))
        else:
            raise TypeError("image must be uint8 or float32")
        self._keep = img

    def set_input_dev(self, dev_ptr, w, h, is_float):
        self._chk(lib().psx_set_input_dev(self._h, C.c_void_p(dev_ptr), w, h, 1 if is_float else 0))

    def set_input_tensor(self, t):
        """t: contiguous 2-D torch tensor (uint8 or float32) already on this context's device."""
        assert t.is_contiguous() and t.dim() == 2
        import torch
        is_float = t.dtype == torch.float32
        assert is_float or t.dtype == torch.uint8
        self._keep = t
        self.set_input_dev(t.data_ptr(), t.shape[1], t.shape[0], is_float)

    # ---- stages ------------------------------------------------------------------------------
    def resize(self, w, h):
        self._chk(lib().psx_resize(self._h, w, h))

    def build_pyramid(self):
        self._chk(lib().psx_build_pyramid(self._h))

    def find_extrema(self):
        self._chk(lib().psx_find_extrema(self._h))

    def orientation(self):
        self._chk(lib().psx_orientation(self._h))

    def descriptors(self):
        self._chk(lib().psx_descriptors(self._h))

    def extract(self):
        self._chk(lib().psx_extract(self._h))

    def sync(self):
        self._chk(lib().psx_sync(self._h))

    # ---- results -----------------------------------------------------------------------------
    @property
    def num_octaves(self):
        return lib().psx_num_octaves(self._h)

    @property
    def num_levels(self):
        return lib().psx_num_levels(self._h)

    def octave_dims(self, o):
        w, h = C.c_int(), C.c_int()
        self._chk(lib().psx_octave_dims(self._h, o, C.byref(w), C.byref(h)))
        return w.value, h.value

    def counts(self):
        ne, no = C.c_int(), C.c_int()
        self._chk(lib().psx_counts(self._h, C.byref(ne), C.byref(no)))
        return ne.value, no.value

    def download(self):
        ne, no = self.counts()
        feats = np.zeros((ne,), dtype=FEATURE_DTYPE)
        desc = np.zeros((no, 128), dtype=np.float32)
        self._chk(lib().psx_download(self._h, feats.ctypes.data_as(C.c_void_p), ne,
                                     desc.ctypes.data_as(C.c_void_p), no))
        return feats, desc

    def clone_results(self, device=0):
        """psx_clone_results into caller-owned device buffers (what FeaturesDev holds), read back:
        (Feature records with device pointers, descriptors, descriptor->extremum map, device address of descriptor 0)."""
        L = lib()
        L.psx_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.psx_dev_free.argtypes = [C.c_int, C.c_void_p]
        L.psx_dev_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.psx_clone_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        ne, no = self.counts()
        pf, pd, pr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        for p, n in ((pf, max(ne, 1) * FEATURE_DEV_DTYPE.itemsize), (pd, max(no, 1) * 512), (pr, max(no, 1) * 4)):
            if L.psx_dev_alloc(device, n, C.byref(p)) != 0:
                raise PopSiftError("psx_dev_alloc failed")
        try:
            self._chk(L.psx_clone_results(self._h, pf, pd, pr))
            feats = np.zeros((ne,), dtype=FEATURE_DEV_DTYPE)
            desc = np.zeros((no, 128), dtype=np.float32)
            rev = np.zeros((no,), dtype=np.int32)
            for arr, p in ((feats, pf), (desc, pd), (rev, pr)):
                if arr.nbytes and L.psx_dev_read(device, arr.ctypes.data_as(C.c_void_p), p, arr.nbytes) != 0:
                    raise PopSiftError("psx_dev_read failed")
            return feats, desc, rev, pd.value
        finally:
            for p in (pf, pd, pr):
                L.psx_dev_free(device, p)

    def attach_export(self, feat_buf, desc_buf):
        """Attach host buffers for zero-copy export.  feat_buf: uint8 buffer (numpy array or pinned
        torch tensor) of n*52 bytes, desc_buf: float32 buffer of m*128 floats.  None detaches."""
        if feat_buf is None:
            self._chk(lib().psx_attach_export(self._h, None, 0, None, 0))
            self._export = None
            return

        def ptr_n(b, itemsize):
            if hasattr(b, "data_ptr"):
                return b.data_ptr(), b.numel() * b.element_size() // itemsize
            return b.ctypes.data, b.nbytes // itemsize
        fp_, fn = ptr_n(feat_buf, FEATURE_DTYPE.itemsize)
        dp_, dn = ptr_n(desc_buf, 128 * 4)
        self._chk(lib().psx_attach_export(self._h, C.c_void_p(fp_), fn, C.c_void_p(dp_), dn))
        self._export = (feat_buf, desc_buf)

    def exported(self):
        """After counts(): numpy views (no copy) of the exported features / descriptors."""
        ne, no = self.counts()
        fb, db = self._export
        fa = fb.numpy() if hasattr(fb, "numpy") else fb
        da = db.numpy() if hasattr(db, "numpy") else db
        feats = fa.view(np.uint8).reshape(-1)[: ne * FEATURE_DTYPE.itemsize].view(FEATURE_DTYPE)
        desc = da.view(np.float32).reshape(-1)[: no * 128].reshape(no, 128)
        return feats, desc

    def dump_plane(self, kind, octave, level):
        w, h = self.octave_dims(octave)
        out = np.zeros((h, w), dtype=np.float32)
        self._chk(lib().psx_dump_plane(self._h, kind, octave, level, out.ctypes.data_as(C.c_void_p)))
        return out

    def dump_iext(self, octave):
        n = C.c_int()
        self._chk(lib().psx_dump_iext(self._h, octave, None, 0, C.byref(n)))
        out = np.zeros((n.value,), dtype=IEXT_DTYPE)
        if n.value:
            self._chk(lib().psx_dump_iext(self._h, octave, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def dump_extrema(self):
        n = C.c_int()
        self._chk(lib().psx_dump_extrema(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value,), dtype=EXT_DTYPE)
        if n.value:
            self._chk(lib().psx_dump_extrema(self._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    # ---- measurement -------------------------------------------------------------------------
    def enable_timers(self, on=True):
        self._chk(lib().psx_enable_timers(self._h, 1 if on else 0))

    def stage_times(self):
        ms = (C.c_float * 4)()
        self._chk(lib().psx_stage_times(self._h, ms))
        return [ms[i] for i in range(4)]

    def time_blur(self, octave, level, reps=20):
        ms, by = C.c_float(), C.c_double()
        self._chk(lib().psx_time_blur(self._h, octave, level, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def enable_blur_probe(self, on=True):
        self._chk(lib().psx_enable_blur_probe(self._h, 1 if on else 0))

    def blur_probe_times(self):
        """Durations (ms) of the octave-0 blur launches of the last extraction, timed in the pipeline, and the
        algorithmic bytes of one launch."""
        ms = (C.c_float * GAUSS_LEVELS)()
        n, by = C.c_int(), C.c_double()
        self._chk(lib().psx_blur_probe_times(self._h, ms, GAUSS_LEVELS, C.byref(n), C.byref(by)))
        return [ms[i] for i in range(n.value)], by.value

    def probe_extra_times(self):
        """(level0 ms, level0 algorithmic bytes, octave-0 extrema ms, its bytes) of the last extraction (blur probe on)."""
        L = lib()
        L.psx_probe_extra_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_double)]
        a, b, c, d = C.c_float(), C.c_double(), C.c_float(), C.c_double()
        self._chk(L.psx_probe_extra_times(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    @property
    def stream(self):
        return lib().psx_stream(self._h)


def copy_bench(device=0, nbytes=0, reps=10):
    """Measured HBM roofline: (GB/s, ms per launch) of the 16 B/lane streaming copy kernel (psx_copy_bench)."""
    ms, by = C.c_float(), C.c_double()
    rc = lib().psx_copy_bench(device, nbytes, reps, C.byref(ms), C.byref(by))
    if rc != 0:
        raise PopSiftError("psx_copy_bench failed (%d)" % rc)
    return by.value / (ms.value * 1e-3) / 1e9, ms.value


# ---- the C++ host library through its flat C binding (include/popsift_c.h) --------------------------
HOST_LIB_PATH = os.environ.get("POPSIFT_HOST_LIB") or os.path.join(_HERE, "lib", "libpopsift.so")
HOST_SYMBOLS = ["popsift_c_create", "popsift_c_destroy", "popsift_c_enqueue_u8", "popsift_c_enqueue_f32",
                "popsift_c_get", "popsift_c_feature_count", "popsift_c_descriptor_count", "popsift_c_copy",
                "popsift_c_descriptors", "popsift_c_free", "popsift_c_last_error", "popsift_c_pool_stats"]
_HOST = None


def host_lib():
    global _HOST
    if _HOST is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise PopSiftError("%s is missing: build it with `python -m popsift_amd.build`" % HOST_LIB_PATH)
        lib()                                   # libpopsift.so links against libpopsift_hip.so
        H = C.CDLL(HOST_LIB_PATH)
        vp = C.c_void_p
        H.popsift_c_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int]
        H.popsift_c_create.restype = vp
        H.popsift_c_destroy.argtypes = [vp]
        H.popsift_c_destroy.restype = None
        for n in ("popsift_c_enqueue_u8", "popsift_c_enqueue_f32"):
            getattr(H, n).argtypes = [vp, C.c_int, C.c_int, vp]
            getattr(H, n).restype = vp
        H.popsift_c_get.argtypes = [vp]
        H.popsift_c_get.restype = vp
        H.popsift_c_feature_count.argtypes = [vp]
        H.popsift_c_descriptor_count.argtypes = [vp]
        H.popsift_c_copy.argtypes = [vp, vp, vp]
        H.popsift_c_descriptors.argtypes = [vp]
        H.popsift_c_descriptors.restype = vp
        H.popsift_c_free.argtypes = [vp]
        H.popsift_c_free.restype = None
        H.popsift_c_last_error.restype = C.c_char_p
        H.popsift_c_pool_stats.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
        H.popsift_c_pool_stats.restype = None
        _HOST = H
    return _HOST


def pool_stats(device=-1):
    """Counters of libpopsift's pinned pool (popsift_c_pool_stats): dict(allocs, frees, hits, free_buffers, free_bytes, in_use)."""
    out = (C.c_longlong * 6)()
    host_lib().popsift_c_pool_stats(device, out)
    return dict(zip(("allocs", "frees", "hits", "free_buffers", "free_bytes", "in_use"), [int(v) for v in out]))


class PopSift:
    """PopSift / SiftJob / FeaturesHost of the C++ library (popsift/popsift.h), through popsift_c.h:
    enqueue(img) -> job handle, get(job) -> (features, descriptors) numpy arrays or just the counts."""

    def __init__(self, cfg=None, device=0, float_images=False):
        self.cfg = cfg if cfg is not None else default_config()
        self._h = host_lib().popsift_c_create(C.byref(self.cfg), 1 if float_images else 0, device)
        if not self._h:
            raise PopSiftError("popsift_c_create failed: %s" % host_lib().popsift_c_last_error().decode())
        self._float = float_images

    def enqueue(self, img):
        """img: C-contiguous (h, w) numpy array, uint8 or float32 (matching the image mode)."""
        h, w = img.shape
        f = host_lib().popsift_c_enqueue_f32 if self._float else host_lib().popsift_c_enqueue_u8
        job = f(self._h, w, h, img.ctypes.data)
        if not job:
            raise PopSiftError("enqueue refused the image: %s" % host_lib().popsift_c_last_error().decode())
        return job

    def get_counts(self, job):
        """SiftJob::get, then only the two counts; the FeaturesHost is deleted."""
        H = host_lib()
        f = H.popsift_c_get(job)
        if not f:
            raise PopSiftError("SiftJob::get failed: %s" % H.popsift_c_last_error().decode())
        n = (H.popsift_c_feature_count(f), H.popsift_c_descriptor_count(f))
        H.popsift_c_free(f)
        return n

    def get(self, job):
        H = host_lib()
        f = H.popsift_c_get(job)
        if not f:
            raise PopSiftError("SiftJob::get failed: %s" % H.popsift_c_last_error().decode())
        ne, no = H.popsift_c_feature_count(f), H.popsift_c_descriptor_count(f)
        feats = np.zeros((ne,), dtype=FEATURE_DTYPE)
        desc = np.zeros((no, 128), dtype=np.float32)
        H.popsift_c_copy(f, feats.ctypes.data, desc.ctypes.data)
        H.popsift_c_free(f)
        return feats, desc

    def close(self):
        if self._h:
            host_lib().popsift_c_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
