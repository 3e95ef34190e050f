#!/usr/bin/env python3
"""Generate tests/golden/sens_*.npz: the REFERENCE's own sources (oracle/_ref, CPU emulation) in two sensitivity builds, on
the frames of ref_config1_vlfeat_640x480.npz and ref_adv_composite_240x160.npz:
    nofma   device code compiled with -ffp-contract=off       (make -C oracle ref_nofma)
    model   CUDA's documented math errors at every stand-in    (make -C oracle ref_model; OSIFT_CUDA_MODEL = rand:1, plus)
What is stored: the variant's Feature records, its descriptors, the SHA-1 of every Gaussian plane.  One variant per
process (the error model is read once per process):
    python tests/golden/make_sensitivity.py nofma | model_rand1 | model_plus
Each run takes several minutes (the fiber emulation).  tests/test_oracle_cpu.py::test_sensitivity_to_cuda_fast_math_models
compares the files with the standard fixtures and with the live oracle."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

VARIANTS = {"nofma": ("libpopsift_ref_nofma.so", None), "model_rand1": ("libpopsift_ref_model.so", "rand:1"),
            "model_plus": ("libpopsift_ref_model.so", "plus")}
FRAMES = ["config1_vlfeat_640x480", "adv_composite_240x160"]


def main():
    v = sys.argv[1]
    so, model = VARIANTS[v]
    os.environ["OSIFT_REF_LIB"] = os.path.join(ROOT, "oracle", "_ref", so)
    if model:
        os.environ["OSIFT_CUDA_MODEL"] = model
    from oracle import pyoracle as po, pyref as pr
    for frame in FRAMES:
        z = np.load(os.path.join(HERE, "ref_%s.npz" % frame), allow_pickle=False)
        kw = json.loads(str(z["config"]))
        r = pr.run(po.default_config(**kw), z["image"])
        sha = []
        for o in range(r.num_octaves):
            for l in range(r.num_levels):
                sha.append(hashlib.sha1(np.ascontiguousarray(r.gauss(o, l)).tobytes()).hexdigest())
        np.savez_compressed(os.path.join(HERE, "sens_%s_%s.npz" % (v, frame)), config=np.array(json.dumps(kw)),
                            variant=np.array(json.dumps({"lib": so, "OSIFT_CUDA_MODEL": model})),
                            plane_sha1=np.array(json.dumps(sha)), features=r.features(), descriptors=r.descriptors())
        print(v, frame, r.ext_total, r.ori_total, flush=True)


if __name__ == "__main__":
    main()
