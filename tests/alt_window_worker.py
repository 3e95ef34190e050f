"""Helper of tests/test_gpu_modes.py::test_alt_descriptor_window_equals_plane: prints, per (configuration, DescMode), an
order-independent digest of the descriptors.  Run once with POPSIFT_ALT_WINDOW=1 and once with =0 (the switch is read once
per process): the LDS window of k_descriptors_alt must give the bits of the plane in HBM."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi                      # noqa: E402
from popsift_amd.synth import synth               # noqa: E402

CASES = [
    (300, 200, 7, dict(octaves=5, sift_mode=2)),                             # windows hang over every border
    (333, 251, 9, dict(octaves=4, levels=5, sigma=2.0)),                     # in-octave sigma 2.3 .. 4.3 and more: windows up to sigma 3.63 fit, the others read the plane
    (160, 120, 3, dict(octaves=3, upscale_factor=0.0, norm_mode=1)),         # no upscaling, L2 normalisation
]


def main():
    out = []
    for (w, h, seed, kw) in CASES:
        img = synth(w, h, seed)
        for mode in (1, 2, 3, 4):
            ctx = capi.Context(capi.default_config(desc_mode=mode, **kw))
            ctx.upload(img)
            ctx.extract()
            f, d = ctx.download()
            rows = sorted(hashlib.sha1(np.ascontiguousarray(r).tobytes()).digest() for r in np.asarray(d))
            out.append(dict(case=[w, h, seed], mode=mode, n=int(len(d)),
                            digest=hashlib.sha1(b"".join(rows)).hexdigest()))
            ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
