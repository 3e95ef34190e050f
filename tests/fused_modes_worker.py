"""Worker of tests/test_gpu_modes.py::test_fused_mode_kernels_equal_per_level_kernels: one process = one setting of the
environment switches (they are read once per process).  Prints one JSON line: for every case a SHA-1 over all Gaussian
planes and the keypoint count."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi                      # noqa: E402
from popsift_amd.synth import synth, synth_float  # noqa: E402

CASES = [
    ("fixed9_1080p", dict(gauss_mode=4), (1920, 1080), False),
    ("fixed15_1080p", dict(gauss_mode=5), (1920, 1080), False),
    ("fixed9_float_odd", dict(gauss_mode=4), (1171, 653), True),
    ("fixed15_odd", dict(gauss_mode=5), (1171, 653), False),
    ("fixed9_direct", dict(gauss_mode=4, scaling_mode=0), (640, 480), False),
    ("fixed15_up0", dict(gauss_mode=5, upscale_factor=0.0), (640, 480), False),
    ("relative_1080p", dict(gauss_mode=1), (1920, 1080), False),
    # upsampled planes 4200 x 2600: the columns / rows around 2048 and 4096 are where c -+ off changes its binade
    ("relative_float_wide", dict(gauss_mode=1), (2100, 1300), True),
    ("relative_opencv_mode", dict(gauss_mode=1, sift_mode=1), (1171, 653), False),
    ("relative_levels4_sigma2", dict(gauss_mode=1, levels=4, sigma=2.0), (800, 600), False),
    ("relative_direct", dict(gauss_mode=1, scaling_mode=0), (640, 480), False),
    ("relative_up0", dict(gauss_mode=1, upscale_factor=0.0), (1171, 653), False),
    ("relative_tiny", dict(gauss_mode=1), (70, 50), False),
    # planes smaller than a strip, a step, a filter: every workgroup is an edge strip / a first-and-last chunk
    ("fixed9_9x7", dict(gauss_mode=4, octaves=2), (9, 7), False),
    ("fixed15_130x5", dict(gauss_mode=5, octaves=2), (130, 5), False),
    ("fixed9_129x65_float", dict(gauss_mode=4, octaves=3), (129, 65), True),
    ("relative_5x130", dict(gauss_mode=1, octaves=2), (5, 130), False),
    ("relative_67x33", dict(gauss_mode=1, octaves=3), (67, 33), True),
    ("relative_4x4", dict(gauss_mode=1, octaves=1), (4, 4), False),
]

out = []
for name, kw, (w, h), is_float in CASES:
    img = synth_float(w, h, 5) if is_float else synth(w, h, 5)
    ctx = capi.Context(capi.default_config(**dict(dict(octaves=5), **kw)))
    ctx.upload(img)
    ctx.extract()
    sha = hashlib.sha1()
    for o in range(ctx.num_octaves):
        for l in range(ctx.num_levels):
            sha.update(np.ascontiguousarray(ctx.dump_plane(capi.PLANE_GAUSS, o, l)).tobytes())
    n = len(ctx.download()[0])
    ctx.close()
    out.append({"case": name, "planes": sha.hexdigest(), "n": int(n)})
print(json.dumps(out))
