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


def big_cases():
    return sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "bigref_*.npz")))


def load_big(name):
    """Full-size fixtures (make_golden.py BIG_CASES): the image is regenerated from its seed (its SHA-1 is checked),
    every plane is a SHA-1, all Feature records are kept, and every desc_stride-th descriptor."""
    from popsift_amd.synth import synth
    z = np.load(os.path.join(GOLDEN_DIR, "bigref_%s.npz" % name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["config"] = json.loads(str(d["config"]))
    d["plane_sha1"] = json.loads(str(d["plane_sha1"]))
    d["dims"] = [tuple(int(v) for v in row) for row in d["dims"]]
    d["num_levels"] = int(d["num_levels"])
    w, h = (int(v) for v in d["size"])
    d["image"] = synth(w, h, int(d["seed"]))
    assert hashlib.sha1(d["image"].tobytes()).hexdigest() == str(d["image_sha1"]), "synth() no longer reproduces the fixture's image"
    d["desc_stride"] = int(d["desc_stride"])
    d["desc_count"] = int(d["desc_count"])
    return d
