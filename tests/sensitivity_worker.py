"""Subprocess side of tests/test_oracle_cpu.py::test_sensitivity_to_cuda_fast_math_models: runs the oracle (OSIFT_LIB) or
the CPU build of the reference's own sources (OSIFT_REF_LIB) -- the environment selects variant builds and OSIFT_CUDA_MODEL
the error model -- on the images of an .npz and saves features, descriptors and per-plane SHA-1s.
  python tests/sensitivity_worker.py {oracle|ref} in.npz out.npz"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    which, src, dst = sys.argv[1:4]
    from oracle import pyoracle as po
    if which == "ref":
        from oracle import pyref as eng
    else:
        eng = po
    z = np.load(src, allow_pickle=False)
    cfgs = json.loads(str(z["__cfgs__"]))
    out = {}
    for name in z.files:
        if name == "__cfgs__":
            continue
        r = eng.run(po.default_config(**cfgs[name]), z[name])
        out["f_" + name] = r.features().copy()
        out["d_" + name] = r.descriptors().copy()
        h = hashlib.sha1()
        for o in range(r.num_octaves):
            for l in range(r.num_levels):
                h.update(np.ascontiguousarray(r.gauss(o, l)).tobytes())
        out["p_" + name] = np.frombuffer(h.digest(), np.uint8).copy()
    np.savez(dst, **out)


if __name__ == "__main__":
    main()
