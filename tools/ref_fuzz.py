#!/usr/bin/env python3
"""Wide random sweep of the oracle against the reference's own code on the CPU (oracle/_ref), run offline in the build
container (needs oracle/_ref/libpopsift_ref.so, i.e. /root/reference at build time).

tests/test_ref_shim_cpu.py holds the cases that run in the suite; this walks the whole Config space -- odd image
sizes, fractional scale factors, every SiftMode / GaussMode / ScalingMode / NormMode, grid filter -- and prints every
case where a Gaussian plane, an extremum count or a feature differs.  Round 4 found the level-0 tap-coordinate
rounding at fractional scale factors with it.

    python tools/ref_fuzz.py [cases] [seed] [max_side]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import pyoracle as oracle, pyref as ref          # noqa: E402  (test infrastructure, not product)
from popsift_amd.synth import synth, synth_float             # noqa: E402
from tests.parity import match_features                      # noqa: E402


def cases(n, seed, max_side):
    rng = np.random.default_rng(seed)
    for i in range(n):
        w, h = int(rng.integers(36, max_side)), int(rng.integers(36, max(48, max_side * 3 // 4)))
        gm = int(rng.choice([0, 0, 1, 2, 3, 4, 5]))
        kw = dict(octaves=int(rng.integers(1, 5)), sift_mode=int(rng.integers(0, 3)), gauss_mode=gm,
                  levels=3 if gm in (4, 5) else int(rng.integers(2, 6)),
                  upscale_factor=float(rng.choice([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 1.0, 1.5, 2.0])),
                  scaling_mode=int(rng.choice([1, 1, 0])),
                  norm_mode=int(rng.integers(0, 2)), norm_multi=int(rng.choice([0, 9])),
                  sigma=float(rng.choice([1.2, 1.6, 2.0])), threshold=float(rng.choice([0.02, 0.04, 0.06])),
                  edge_limit=float(rng.choice([8.0, 10.0, 16.0])),
                  initial_blur=float(rng.choice([0.0, 0.5, 0.8])),
                  desc_mode=int(rng.choice([0, 0, 0, 1, 3, 4])))   # grid (2) is a bound, not a match: tests/test_ref_shim_cpu.py
        if kw["upscale_factor"] >= 1.5 and max(w, h) > 160:
            w, h = w // 2 + 20, h // 2 + 20                  # the emulation is slow: keep octave 0 under ~0.3 Mpix
        if rng.random() < 0.25:
            kw.update(filter_max_extrema=int(rng.integers(20, 300)), filter_grid_size=int(rng.integers(1, 4)),
                      grid_filter_mode=int(rng.integers(1, 3)))     # RandomScale depends on buffer order: left out
        yield i, w, h, 9000 + i, bool(rng.random() < 0.3), kw


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    bad = 0
    t0 = time.time()
    for i, w, h, s, is_float, kw in cases(n, seed, max_side):
        img = synth_float(w, h, s) if is_float else synth(w, h, s)
        try:
            cfg = oracle.default_config(**kw)
            r, o = ref.run(cfg, img), oracle.run(cfg, img)
        except Exception as e:                               # both sides must refuse the same configurations
            print("case %d %dx%d float=%d %s: %s: %s" % (i, w, h, is_float, kw, type(e).__name__, e), flush=True)
            continue
        why = []
        if r.dims != o.dims or r.num_levels != o.num_levels:
            why.append("dims %s vs %s" % (r.dims, o.dims))
        else:
            for oc in range(r.num_octaves):
                for l in range(r.num_levels):
                    if not np.array_equal(r.gauss(oc, l).view(np.uint32), o.gauss(oc, l).view(np.uint32)):
                        why.append("plane (%d,%d) max err %g" % (oc, l, float(np.abs(r.gauss(oc, l) - o.gauss(oc, l)).max())))
                        break
                kept = o.iext(oc)                            # the reference compacts what the grid filter drops
                nkept = int((kept["ignore"] == 0).sum()) if "filter_max_extrema" in kw else len(kept)
                if len(r.iext(oc)) != nkept:
                    why.append("octave %d extrema %d vs %d" % (oc, len(r.iext(oc)), nkept))
            if (r.ext_total, r.ori_total) != (o.ext_total, o.ori_total):
                why.append("totals %s vs %s" % ((r.ext_total, r.ori_total), (o.ext_total, o.ori_total)))
            elif r.ext_total and not why:
                m = match_features(r.features(), r.descriptors(), o.features(), o.descriptors(),
                                   norm_scale=float(2 ** kw["norm_multi"]))
                if not (m["kp_match"] == 1.0 and m["ori_match"] == 1.0 and m["desc_match"] == 1.0):
                    why.append("features %s" % m)
        if why:
            bad += 1
            print("MISMATCH case %d %dx%d float=%d %s: %s" % (i, w, h, is_float, kw, "; ".join(why[:4])), flush=True)
        else:
            print("ok case %d %dx%d float=%d kp=%d (%.0f s)" % (i, w, h, is_float, o.ext_total, time.time() - t0), flush=True)
    print("%d cases, %d mismatching" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
