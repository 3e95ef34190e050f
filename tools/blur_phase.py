import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from popsift_amd import capi
from popsift_amd.synth import synth
L = capi.lib()
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img); ctx.extract(); ctx.sync()
buf = torch.zeros(1100 * 8, dtype=torch.int64, device="cuda")
for lvl in (1, 2, 3, 4, 5):
    L.psx_debug_set_blur_buffer(C.c_void_p(buf.data_ptr()))
    buf.zero_(); torch.cuda.synchronize()
    ms, _ = ctx.time_blur(0, lvl, 1)
    b = buf.cpu().numpy().reshape(-1, 8); b = b[b[:, 6] > 0]
    print("level", lvl, "blocks", len(b), "kernel %.1f us" % (ms * 1e3), "mean cycles: commit %.0f bar1 %.0f H %.0f bar2 %.0f V %.0f | total/WG %.0f steps %.1f" % (*b[:, :5].mean(0), b[:, 5].mean(), b[:, 6].mean()))
    L.psx_debug_set_blur_buffer(None)
