"""Load the committed reference fixtures (tests/golden/ref_*.npz, made by tests/golden/make_golden.py
from the reference's own code running on the CPU)."""
import glob
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "ref_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, "ref_%s.npz" % name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["config"] = json.loads(str(d["config"]))
    d["plane_sha1"] = json.loads(str(d["plane_sha1"]))
    d["dims"] = [tuple(int(v) for v in row) for row in d["dims"]]
    d["num_levels"] = int(d["num_levels"])
    return d


def sha1(plane):
    return hashlib.sha1(np.ascontiguousarray(plane, dtype=np.float32).tobytes()).hexdigest()
