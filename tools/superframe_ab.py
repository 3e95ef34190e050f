"""dev tool (GPU box): what would batching frames into larger launches be worth?  The same frame content at 1920x1080
(5 octaves) and tiled to 4096x4096 (6 octaves: 8x the pixels per launch chain), device resident, 1..N contexts in flight.
If the large frames with a few contexts in flight beat the 1080p frames with 16, launches are what limits the latter.
python tools/superframe_ab.py"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from popsift_amd import capi
from popsift_amd.synth import synth

base = synth(1920, 1080, 1000)
cases = [("1080p", base, 5, (1, 2, 4, 8, 16)),
         ("2160p", np.ascontiguousarray(np.tile(base, (2, 2))), 6, (1, 2, 4, 8)),
         ("4096sq", np.ascontiguousarray(np.tile(base, (4, 3))[:4096, :4096]), 6, (1, 2, 3, 4))]
for name, img, octs, ns in cases:
    h, w = img.shape
    t_img = torch.from_numpy(img).cuda()
    for n in ns:
        ctxs = [capi.Context(capi.default_config(octaves=octs, sift_mode=2)) for _ in range(n)]
        for c in ctxs:
            c.set_input_tensor(t_img)
        for _ in range(3):
            for c in ctxs: c.extract()
            for c in ctxs: kp = c.counts()
        reps = max(4, int(200e6 / (w * h) / n))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            for c in ctxs:
                if r: c.counts()
                c.extract()
        for c in ctxs: c.counts()
        dt = time.perf_counter() - t0
        print(json.dumps({"frame": name, "contexts": n, "mpix_s": round(w * h * n * reps / dt / 1e6, 1),
                          "ms_per_frame": round(dt / (n * reps) * 1e3, 4), "keypoints": kp[0], "kp_per_1000px": round(kp[0] * 1e3 / (w * h), 2)}), flush=True)
        for c in ctxs: c.close()
