#!/usr/bin/env python3
"""A/B of pyramid schedules / kernel switches on the GPU, one subprocess per environment so that every POPSIFT_* switch is
read fresh (they are read at psx_create or once per process).  Per variant: stage times of one frame on one context (HIP
events), median wall time of one frame, in-pipeline k_blur probe, device-resident throughput over NCTX contexts.

  python tools/sched_ab.py [--size W H] [--repeat N] 'POPSIFT_TILE=0' 'POPSIFT_TILE=1 POPSIFT_TILE_TY=32' ...

Every positional argument is one variant: space-separated NAME=VALUE pairs ('' = the defaults).  Variants run in the given
order, the whole list --repeat times (boxes drift: compare neighbours, and list the baseline first and last).
Valid switches are listed in tools/README.md; psx_create ignores values it does not know (e.g. POPSIFT_FLOW accepts 0..2)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(w, h):
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
    nf = 600 if w < 4096 else 30
    best = None
    for rep in range(3):
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
        best = dt if best is None or dt < best else best
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("POPSIFT_")}, "size": [w, h],
                      "keypoints": n[0], "single_ms": round(lat[len(lat) // 2] * 1e3, 4), "single_min_ms": round(lat[0] * 1e3, 4),
                      "stage_ms": [round(v, 4) for v in st], "probe_ms": [round(v, 5) for v in pm], "probe_bytes": pb,
                      "throughput_mpix": round(nf * w * h / best / 1e6, 1), "nctx": nctx}), flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "worker":
        worker(int(args[1]), int(args[2]))
        return
    w, h, repeat = 1920, 1080, 1
    variants = []
    i = 0
    while i < len(args):
        if args[i] == "--size":
            w, h = int(args[i + 1]), int(args[i + 2]); i += 3
        elif args[i] == "--repeat":
            repeat = int(args[i + 1]); i += 2
        else:
            variants.append(dict(kv.split("=", 1) for kv in args[i].split())); i += 1
    if not variants:
        variants = [{}]
    for _ in range(repeat):
        for v in variants:
            e = {k: x for k, x in os.environ.items() if not k.startswith("POPSIFT_")}
            e.update(v)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", str(w), str(h)], env=e, cwd=ROOT,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            print(p.stdout.strip() if p.returncode == 0 else json.dumps({"env": v, "failed": p.stderr[-800:]}), flush=True)


if __name__ == "__main__":
    main()
