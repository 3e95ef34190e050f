"""In-pipeline durations of the octave-0 blur launches (one context, 1080p bench frame) under the current
POPSIFT_BLUR_* environment; one JSON line.  Used to A/B kernel variants in one gpurun call."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import capi
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
rng = np.random.default_rng(1)
img = (rng.random((h // 8 + 1, w // 8 + 1)) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w].copy()
ctx = capi.Context(capi.default_config(octaves=5))
ctx.upload(img)
ctx.enable_blur_probe(True)
acc, n = None, 0
for it in range(30):
    ctx.extract()
    ms, by = ctx.blur_probe_times()
    if it >= 6:
        acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
        n += 1
per = [a / n * 1e3 for a in acc]
avg = sum(per) / len(per)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("POPSIFT_")}, "per_level_us": [round(p, 2) for p in per],
                  "avg_us": round(avg, 2), "GBs": round(by / (avg * 1e-6) / 1e9, 1), "frac_of_8TBs": round(by / (avg * 1e-6) / 8e12, 4)}))
ctx.close()
