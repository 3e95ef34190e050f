#!/usr/bin/env python3
"""Level 0 of octave 0 on the GPU: k_level0_x2 (POPSIFT_LEVEL0_X2 unset) against round 3's k_level0_fused (=0), timed in
the pipeline with the probe's stream events; u8 and float input, shift 1.0 (VLFeat) and 0.5 (OpenCV mode).
  python tools/level0_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import numpy as np
    import torch
    from popsift_amd import capi
    from popsift_amd.synth import synth
    dev = torch.device("cuda", 0)
    img = synth(1920, 1080, 1000)
    out = {}
    for name, kw, fl in (("u8_vlfeat", dict(octaves=5, sift_mode=2), False), ("f32_popsift", dict(octaves=5), True),
                         ("u8_opencv", dict(octaves=5, sift_mode=1), False)):
        t = torch.from_numpy((img.astype(np.float32) / 256.0) if fl else img).to(dev)
        c = capi.Context(capi.default_config(**kw))
        c.set_input_tensor(t)
        c.enable_blur_probe(True)
        acc = []
        for i in range(30):
            c.extract()
            c.blur_probe_times()
            if i >= 6:
                acc.append(c.probe_extra_times()[0])
        c.enable_blur_probe(False)
        c.enable_timers(True)
        st = []
        for i in range(9):
            c.extract(); st.append(c.stage_times()[0])
        out[name] = {"level0_us": round(1e3 * sorted(acc)[len(acc) // 2], 2), "pyramid_stage_us": round(1e3 * sorted(st)[4], 1)}
        c.close()
    print(json.dumps({"x2": os.environ.get("POPSIFT_LEVEL0_X2", "1"), **out}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker()
    else:
        for v in ("0", "1", "0", "1"):
            e = dict(os.environ, POPSIFT_LEVEL0_X2=v)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "w"], env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            print(p.stdout.strip() or p.stderr[-500:], flush=True)
