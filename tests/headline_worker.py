"""Runs the end-to-end leg of bench.py (host frames -> PopSift::enqueue -> SiftJob::get -> FeaturesHost) over
bench.py's own frames in a separate process and saves every result -- for the settings that are read once per
process (POPSIFT_EXPORT, POPSIFT_PINNED_LIMIT_MB).  Used by tests/test_gpu_headline.py.

  python -m tests.headline_worker OUT.npz NFRAMES OUTSTANDING PASSES [keep]

`keep`: hold every FeaturesHost alive until the end (a caller that hoards results)."""
import os
import sys
from collections import deque

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def stream(ps, frames, outstanding, passes=1, get=None):
    """The loop of bench.py's e2e_step / e2e_drain with full results: `outstanding` jobs in flight, FIFO."""
    get = get or ps.get
    jobs, res = deque(), []
    for _ in range(passes):
        for f in frames:
            if len(jobs) >= outstanding:
                res.append(get(jobs.popleft()))
            jobs.append(ps.enqueue(f))
    while jobs:
        res.append(get(jobs.popleft()))
    return res


def bench_frames(n):
    import bench
    from popsift_amd.synth import synth
    return bench.make_frames(0, 1, synth)[:n]


def main():
    out, n, outstanding, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    keep = len(sys.argv) > 5 and sys.argv[5] == "keep"
    from popsift_amd import capi
    frames = bench_frames(n)
    ps = capi.PopSift(capi.default_config(octaves=5))
    held = []
    if keep:
        H = capi.host_lib()

        def get(job):
            f = H.popsift_c_get(job)
            if not f:
                raise RuntimeError(H.popsift_c_last_error().decode())
            held.append(f)                          # not freed: the pinned pool runs dry
            ne, no = H.popsift_c_feature_count(f), H.popsift_c_descriptor_count(f)
            feats = np.zeros((ne,), dtype=capi.FEATURE_DTYPE)
            desc = np.zeros((no, 128), dtype=np.float32)
            H.popsift_c_copy(f, feats.ctypes.data, desc.ctypes.data)
            return feats, desc
    else:
        get = None
    res = stream(ps, frames, outstanding, passes, get)
    for f in held:
        capi.host_lib().popsift_c_free(f)
    ps.close()
    np.savez(out, **{"f%d" % i: r[0] for i, r in enumerate(res)}, **{"d%d" % i: r[1] for i, r in enumerate(res)})


if __name__ == "__main__":
    main()
