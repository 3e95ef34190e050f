#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz from the REFERENCE's own code (oracle/_ref/libpopsift_ref.so:
/root/reference/src/popsift compiled for the CPU through the CUDA emulation in oracle/ref_shim).

Run in the build container (where /root/reference exists):
    make -C oracle ref && python tests/golden/make_golden.py
The fixtures are small (a few tens of kB each) and are committed; the GPU box has no reference.
Each fixture holds the input image, the config overrides, every Gaussian plane's SHA-1, the
initial extrema per octave, the Feature records and the descriptors as produced by the reference.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po, pyref as pr   # noqa: E402
from popsift_amd.synth import synth, synth_float  # noqa: E402
from tests import adversarial as adv  # noqa: E402

CASES = {
    # name: (w, h, seed, float_input, config overrides)
    "popsift_96x72": (96, 72, 101, False, dict(octaves=3)),
    "vlfeat_112x80": (112, 80, 102, False, dict(octaves=3, sift_mode=po.MODE_VLFEAT)),
    "opencv_96x80": (96, 80, 103, False, dict(octaves=3, sift_mode=po.MODE_OPENCV, gauss_mode=po.GAUSS_OPENCV_COMPUTE)),
    "float_up0_classic_160x120": (160, 120, 104, True, dict(octaves=3, upscale_factor=0.0, norm_mode=po.NORM_CLASSIC, norm_multi=9)),
    "auto_octaves_75x61": (75, 61, 105, False, dict()),
    # a larger frame with four octaves in the mode the north_star quotes parity on (VLFeat), RootSift x512
    "vlfeat_256x192_oct4": (256, 192, 106, False, dict(octaves=4, sift_mode=po.MODE_VLFEAT, norm_multi=9)),
    # extrema_filter_grid (s_filtergrid.cu, compiled against oracle/ref_shim/thrust_shim.h); the
    # RandomScale mode depends on buffer order and has no order-free golden answer beyond counts
    "gridfilter_largest_160x120": (160, 120, 99, False, dict(octaves=3, filter_max_extrema=38, filter_grid_size=2,
                                                             grid_filter_mode=po.FILTER_LARGEST_FIRST)),
    "gridfilter_smallest_g3_160x120": (160, 120, 99, False, dict(octaves=3, filter_max_extrema=57, filter_grid_size=3,
                                                                 grid_filter_mode=po.FILTER_SMALLEST_FIRST)),
    # BASELINE config 1 at full size: 640x480, 5 octaves, VLFeat mode (a few minutes in the fiber emulation)
    "config1_vlfeat_640x480": (640, 480, 1000, False, dict(octaves=5, sift_mode=po.MODE_VLFEAT)),
    # alternative pyramid / descriptor modes, tests/test_gpu_modes.py
    "mode_relative_120x90": (120, 90, 201, False, dict(octaves=3, gauss_mode=po.GAUSS_VLFEAT_RELATIVE)),
    "mode_relative_all_120x90": (120, 90, 202, False, dict(octaves=3, gauss_mode=po.GAUSS_VLFEAT_RELATIVE_ALL)),
    "mode_fixed9_120x90": (120, 90, 203, False, dict(octaves=3, gauss_mode=po.GAUSS_FIXED9)),
    "mode_fixed15_direct_120x90": (120, 90, 204, True, dict(octaves=3, gauss_mode=po.GAUSS_FIXED15, scaling_mode=po.SCALE_DIRECT)),
    "mode_direct_relative_120x90": (120, 90, 205, False, dict(octaves=3, gauss_mode=po.GAUSS_VLFEAT_RELATIVE, scaling_mode=po.SCALE_DIRECT)),
    "mode_iloop_120x90": (120, 90, 206, False, dict(octaves=3, desc_mode=po.DESC_ILOOP)),
    "mode_grid_120x90": (120, 90, 207, False, dict(octaves=3, desc_mode=po.DESC_GRID)),
    "mode_igrid_120x90": (120, 90, 208, False, dict(octaves=3, desc_mode=po.DESC_IGRID, norm_mode=po.NORM_CLASSIC, norm_multi=9)),
    "mode_notile_120x90": (120, 90, 209, False, dict(octaves=3, desc_mode=po.DESC_NOTILE)),
    # adversarial content (tests/adversarial.py): saturated plateaus, step edges, binary strokes, exact DoG ties, a
    # mosaic of all of them, and the two images without any extremum (sine grating, pure ramp); round 3
    "adv_plateaus_192x144": (192, 144, "plateaus", False, dict(octaves=4)),
    "adv_checker_160x120": (160, 120, "checker", False, dict(octaves=3, sift_mode=po.MODE_VLFEAT)),
    "adv_textlike_160x120": (160, 120, "textlike", False, dict(octaves=3)),
    "adv_stripes_160x120": (160, 120, "stripes", False, dict(octaves=3, sift_mode=po.MODE_OPENCV, gauss_mode=po.GAUSS_OPENCV_COMPUTE)),
    "adv_composite_240x160": (240, 160, "composite", False, dict(octaves=4)),
    "adv_grating_128x96": (128, 96, "grating", False, dict(octaves=3)),
    "adv_ramp_128x96": (128, 96, "ramp", False, dict(octaves=3)),
    # scale factors other than -1 / 0 / +1 (popsift.cpp:109-126, s_pyramid_build.cu:96-126): x4 upsampling, x4
    # downsampling and a fractional factor (k_upscale + k_blur<R, true> path of the HIP side); round 4
    "up2_vlfeat_72x56": (72, 56, 301, False, dict(octaves=3, upscale_factor=2.0, sift_mode=po.MODE_VLFEAT)),
    "down2_200x152": (200, 152, 302, False, dict(octaves=2, upscale_factor=-2.0)),
    "up_half_opencv_120x90": (120, 90, 303, False, dict(octaves=3, upscale_factor=0.5, sift_mode=po.MODE_OPENCV)),
    "up_1p5_float_96x72": (96, 72, 304, True, dict(octaves=3, upscale_factor=1.5, sift_mode=po.MODE_VLFEAT)),
    # sizes at which a tap coordinate sits on a 1/256 sub-texel boundary (round 4, found by tools/ref_fuzz.py): level 0 of
    # a fractional scale factor -- (x + shift)/W -+ k/W is not (x -+ k + shift)/W there -- and the fixed-span taps
    # ypos -+ i * mul_h, one fma in the reference's device code
    "up_half_boundary_163x122": (163, 122, 11, False, dict(octaves=2, upscale_factor=0.5, sift_mode=po.MODE_OPENCV)),
    "fixed15_up_half_136x135": (136, 135, 9080, False, dict(octaves=1, sift_mode=po.MODE_OPENCV, gauss_mode=po.GAUSS_FIXED15,
                                                            upscale_factor=0.5, initial_blur=0.8, threshold=0.02)),
}

# BASELINE config 2 at full size in the mode the north_star quotes parity on: the bench frame (seed 1000), VLFeat
# mode, through the reference's own code (about half an hour in the fiber emulation).  Too large to store whole:
# the image is regenerated from its seed, every plane is a SHA-1, all Feature records are kept and every
# DESC_STRIDE-th descriptor.
BIG_CASES = {
    "config2_vlfeat_1920x1080": (1920, 1080, 1000, dict(octaves=5, sift_mode=po.MODE_VLFEAT)),
}
DESC_STRIDE = 8


def make_big(out_dir, only):
    for name, (w, h, seed, kw) in BIG_CASES.items():
        path = os.path.join(out_dir, "bigref_%s.npz" % name)      # not ref_*: its layout differs (golden_util.load_big)
        if name not in only:
            continue                                   # only on request: python make_golden.py config2_vlfeat_1920x1080
        img = synth(w, h, seed)
        r = pr.run(po.default_config(**kw), img)
        planes = {}
        for o in range(r.num_octaves):
            for l in range(r.num_levels):
                planes["g_%d_%d" % (o, l)] = hashlib.sha1(np.ascontiguousarray(r.gauss(o, l)).tobytes()).hexdigest()
        d = r.descriptors()
        data = dict(seed=seed, size=np.array([w, h], np.int32), image_sha1=hashlib.sha1(img.tobytes()).hexdigest(),
                    config=json.dumps(kw), dims=np.array(r.dims, dtype=np.int32), num_levels=r.num_levels,
                    plane_sha1=json.dumps(planes), features=r.features(), desc_stride=DESC_STRIDE,
                    desc_count=len(d), descriptors_sub=d[::DESC_STRIDE].copy())
        for o in range(r.num_octaves):
            data["iext_%d" % o] = r.iext(o)
        np.savez_compressed(path, **data)
        print(name, "octaves", r.num_octaves, "features", r.ext_total, "descriptors", r.ori_total)


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])
    make_big(out_dir, only)
    for name, (w, h, seed, is_float, kw) in CASES.items():
        path = os.path.join(out_dir, "ref_%s.npz" % name)
        if (only and name not in only) or (not only and os.path.exists(path)):
            continue                                   # existing fixtures are kept byte for byte; name them to redo
        if isinstance(seed, str):
            img = adv.make(seed, w, h)                 # named adversarial content instead of a synth() seed
        else:
            img = synth_float(w, h, seed) if is_float else synth(w, h, seed)
        cfg = po.default_config(**kw)
        r = pr.run(cfg, img)
        planes = {}
        for o in range(r.num_octaves):
            for l in range(r.num_levels):
                planes["g_%d_%d" % (o, l)] = hashlib.sha1(np.ascontiguousarray(r.gauss(o, l)).tobytes()).hexdigest()
        data = dict(
            image=img, config=json.dumps(kw), dims=np.array(r.dims, dtype=np.int32),
            num_levels=r.num_levels, plane_sha1=json.dumps(planes),
            gauss_last=r.gauss(r.num_octaves - 1, r.num_levels - 1),
            features=r.features(), descriptors=r.descriptors(),
        )
        for o in range(r.num_octaves):
            data["iext_%d" % o] = r.iext(o)      # the extrema that reach orientation (grid-filter survivors)
        # the public API path (PopSift::enqueue / SiftJob::get) must agree with the direct drive
        ra = pr.run(cfg, img, api=True)
        assert ra.ext_total == r.ext_total and ra.ori_total == r.ori_total
        np.savez_compressed(path, **data)
        print(name, "octaves", r.num_octaves, "features", r.ext_total, "descriptors", r.ori_total)


if __name__ == "__main__":
    main()
