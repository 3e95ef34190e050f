"""psx_match with and without the MFMA prefilter (one subprocess per setting: the switch is read once per process):
seconds and G pairs/s (device-resident descriptors in, results in host memory out), results compared by SHA-1.
  python tools/match_ab.py [n]     n unit-norm random descriptors per side (default 18432);
                                   n = 0: the real descriptors of two 1080p bench frames"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(n):
    import numpy as np
    from popsift_amd import capi
    rng = np.random.default_rng(5)

    def unit(k):
        v = rng.random((k, 128), dtype=np.float32) ** 4
        return np.sqrt(v / v.sum(1, keepdims=True)).astype(np.float32)

    if n == 0:
        from popsift_amd.synth import synth
        a = synth(1920, 1080, 1000)
        out = []
        for img in (a, np.roll(a, 3, axis=1)):
            ctx = capi.Context(capi.default_config(octaves=5, sift_mode=2))
            ctx.upload(img); ctx.extract(); out.append(ctx.download()[1]); ctx.close()
        l, r = out
    else:
        l, r = unit(n), unit(n)
    dl, dr = capi.DeviceDescriptors(l), capi.DeviceDescriptors(r)
    dl.match(dr)
    best = None
    for _ in range(5):
        t = time.perf_counter(); mm, dd = dl.match(dr); dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    print(json.dumps({"mfma": os.environ.get("POPSIFT_MATCH_MFMA", "1"), "left": len(l), "right": len(r), "seconds": round(best, 6),
                      "gpairs_per_s": round(len(l) * len(r) / best / 1e9, 1),
                      "sha1": hashlib.sha1(mm.tobytes() + dd.tobytes()).hexdigest(), "accepted": int((mm[:, 2] == 1).sum())}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(int(sys.argv[2]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 18432
        for v in ("0", "1"):
            e = dict(os.environ, POPSIFT_MATCH_MFMA=v)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", str(n)], env=e, cwd=ROOT,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print(p.stdout.strip()[-1500:])
