"""Worker of tests/test_gpu_parity.py::test_descriptor_lds_layouts_give_identical_descriptors: one process = one setting of
POPSIFT_DESC_OCC (read once per process).  Prints one JSON line: for every case the number of descriptors and a
SHA-1 over (feature record, descriptor) rows sorted by the record -- the order of the extrema list is not deterministic, its
content is."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi                      # noqa: E402
from popsift_amd.synth import synth, synth_float  # noqa: E402

CASES = [
    ("1080p", dict(), (1920, 1080), False),
    ("1080p_popsift_mode", dict(sift_mode=0), (1920, 1080), False),
    ("odd_float", dict(), (1171, 653), True),
    ("up0", dict(upscale_factor=0.0), (1171, 653), False),
    # large sigma: windows taller than 64 rows (two blocks of row spans) and wider than 128 pixels
    ("sigma2_levels2", dict(sigma=2.0, levels=2), (1280, 720), False),
    ("levels5", dict(levels=5), (800, 600), False),
    ("down1", dict(upscale_factor=-1.0), (1920, 1080), False),
    ("classic_norm_multi9", dict(norm_mode=1, norm_multi=9), (800, 600), False),
    ("tiny", dict(octaves=2), (70, 50), False),
]

out = []
for name, kw, (w, h), is_float in CASES:
    img = synth_float(w, h, 9) if is_float else synth(w, h, 9)
    ctx = capi.Context(capi.default_config(**dict(dict(octaves=5), **kw)))
    ctx.upload(img)
    ctx.extract()
    f, d = ctx.download()
    ctx.close()
    f = np.asarray(f)
    d = np.asarray(d)
    rows = []
    # one row per descriptor: the feature's position / scale, the orientation that owns the descriptor, the 128 values
    for i in range(len(f)):
        for k in range(int(f["num_ori"][i])):
            di = int(f["desc_idx"][i][k])
            if di >= 0:
                rows.append(np.concatenate([[f["xpos"][i], f["ypos"][i], f["sigma"][i], f["orientation"][i][k]], d[di]]).astype(np.float32))
    rows = np.array(rows, np.float32).reshape(-1, 132)
    order = np.lexsort(rows.T[::-1])
    sha = hashlib.sha1(np.ascontiguousarray(rows[order]).tobytes()).hexdigest()
    out.append({"case": name, "n": int(len(rows)), "sha": sha})
print(json.dumps(out))
