#!/usr/bin/env python3
"""A/B of the pyramid schedules on the GPU (one subprocess per environment, so that every switch is read fresh):
POPSIFT_FLOW = 0 (one launch per level) / 1 (k_pyramid_flow, every blur level in one launch) / 2 (octave 0 by launches,
octaves >= 1 in one launch), POPSIFT_FLOW_LD = 1 (plain loads) / 2 (sc1 loads), POPSIFT_FLOW_ORDER, POPSIFT_FLOW_GRID.
Per variant: stage times of one frame on one context (HIP events), median wall time of one frame, device-resident
throughput over NCTX contexts.   python tools/flow_ab.py [W H]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(w, h):
    import numpy as np
    import torch
    from popsift_amd import capi
    from popsift_amd.synth import synth
    dev = torch.device("cuda", 0)
    octaves = 6 if w >= 4096 else 5
    cfg = capi.default_config(octaves=octaves, sift_mode=2)
    frames = [torch.from_numpy(synth(w, h, 1000 + i)).to(dev) for i in range(4 if w < 4096 else 1)]
    c0 = capi.Context(cfg)
    c0.set_input_tensor(frames[0])
    for _ in range(5):
        c0.extract(); c0.counts()
    lat = []
    for _ in range(40):
        t = time.perf_counter(); c0.extract(); n = c0.counts(); lat.append(time.perf_counter() - t)
    lat.sort()
    c0.enable_timers(True)
    st = []
    for _ in range(9):
        c0.extract(); st.append(c0.stage_times())
    c0.enable_timers(False)
    st = [sorted(s[i] for s in st)[len(st) // 2] for i in range(4)]
    c0.enable_blur_probe(True)
    probe = []
    for _ in range(6):
        c0.extract(); probe.append(c0.blur_probe_times())
    c0.enable_blur_probe(False)
    pm, pb = probe[-1]
    nctx = 8 if w < 4096 else 2
    ctxs = [c0] + [capi.Context(cfg) for _ in range(nctx - 1)]
    nf = 400 if w < 4096 else 30
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(nf):
            c = ctxs[i % nctx]
            if i >= nctx:
                c.counts()
            c.set_input_tensor(frames[i % len(frames)])
            c.extract()
        for c in ctxs:
            c.counts()
        dt = time.perf_counter() - t0
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("POPSIFT_FLOW")}, "size": [w, h],
                      "keypoints": n[0], "single_ms": round(lat[len(lat) // 2] * 1e3, 4), "single_min_ms": round(lat[0] * 1e3, 4),
                      "stage_ms": [round(v, 4) for v in st], "probe_ms": [round(v, 5) for v in pm], "probe_bytes": pb,
                      "throughput_mpix": round(nf * w * h / dt / 1e6, 1), "nctx": nctx}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(int(sys.argv[2]), int(sys.argv[3]))
        return
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    F = {"POPSIFT_FLOW": "1"}
    variants = [{"POPSIFT_FLOW": "0"}, {"POPSIFT_FLOW": "3", "POPSIFT_FLOW_GRID": "128"}, {"POPSIFT_FLOW": "3", "POPSIFT_FLOW_GRID": "256"},
                {"POPSIFT_FLOW": "3", "POPSIFT_FLOW_GRID": "384"}, {"POPSIFT_FLOW": "3", "POPSIFT_FLOW_GRID": "256", "POPSIFT_FLOW_STEPS": "0,2,1"},
                {"POPSIFT_FLOW": "3", "POPSIFT_FLOW_GRID": "256", "POPSIFT_FLOW_STEPS": "0,3,2"},
                {"POPSIFT_FLOW": "3", "POPSIFT_FLOW_GRID": "192", "POPSIFT_FLOW_STEPS": "0,2,2"}, {"POPSIFT_FLOW": "0"}]
    if os.environ.get("FLOW_AB_ONLY"):
        variants = [v for i, v in enumerate(variants) if str(i) in os.environ["FLOW_AB_ONLY"].split(",")]
    for v in variants:
        e = dict(os.environ); e.update(v)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", str(w), str(h)], env=e, cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        print(p.stdout.strip() if p.returncode == 0 else json.dumps({"env": v, "failed": p.stderr[-800:]}), flush=True)


if __name__ == "__main__":
    main()
