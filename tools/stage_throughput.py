#!/usr/bin/env python3
"""Marginal cost of every stage with frames in flight (GPU): 16 contexts round-robin, device-resident frames, the chain cut
after the pyramid / the extrema / the orientations / the descriptors.  ms per frame of each prefix and the differences:
what a stage costs in THROUGHPUT terms (its kernels overlap other frames' kernels), next to its single-frame duration.
  python tools/stage_throughput.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from popsift_amd import capi
    from popsift_amd.synth import synth
    dev = torch.device("cuda", 0)
    frames = [torch.from_numpy(synth(1920, 1080, 1000 + i)).to(dev) for i in range(8)]
    cfg = capi.default_config(octaves=5, sift_mode=2)
    nctx = 16
    ctxs = [capi.Context(cfg) for _ in range(nctx)]
    for i, c in enumerate(ctxs):
        c.set_input_tensor(frames[i % 8]); c.extract(); c.counts()
    stages = [("pyramid", lambda c: c.build_pyramid()),
              ("+extrema", lambda c: (c.build_pyramid(), c.find_extrema())),
              ("+orientation", lambda c: (c.build_pyramid(), c.find_extrema(), c.orientation())),
              ("+descriptors (all)", lambda c: c.extract())]
    out = {}
    prev = 0.0
    for name, fn in stages:
        best = None
        for rep in range(3):
            n = 480
            t0 = time.perf_counter()
            for i in range(n):
                c = ctxs[i % nctx]
                if i >= nctx:
                    c.sync()
                c.set_input_tensor(frames[i % 8])
                fn(c)
            for c in ctxs:
                c.sync()
            dt = (time.perf_counter() - t0) / n * 1e3
            best = dt if best is None or dt < best else best
        out[name] = {"ms_per_frame": round(best, 4), "marginal_ms": round(best - prev, 4)}
        prev = best
    c0 = ctxs[0]
    c0.enable_timers(True)
    st = []
    for _ in range(9):
        c0.extract(); st.append(c0.stage_times())
    out["single_frame_stage_ms"] = [round(sorted(s[i] for s in st)[4], 4) for i in range(4)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
