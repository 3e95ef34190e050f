"""dev tool: per-phase clock stamps of k_extrema (library must be built with -DPSX_PHASE_TIMING)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from popsift_amd import capi
from popsift_amd.synth import synth
L = capi.lib()
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img); ctx.extract(); ctx.sync()
buf = torch.zeros(8200 * 8, dtype=torch.int64, device="cuda")
L.psx_debug_set_buffer(C.c_void_p(buf.data_ptr()))
ctx.build_pyramid(); ctx.sync()
for o in (4, 2, 0):
    buf.zero_(); torch.cuda.synchronize()
    # run only one octave's extrema: use the private launch through find_extrema on all, stamps of last launched octave win
    ctx.build_pyramid(); ctx.sync()
    L.psx_debug_launch_extrema(ctx._h, o); ctx.sync()
    b = buf.cpu().numpy().reshape(-1, 8)
    b = b[b[:, 0] != 0]
    d = np.diff(b[:, :5], axis=1)
    print("octave", o, "blocks", len(b), "mean phase clocks [param, stage, scan, refine]:", d.mean(0).round(0), "max", d.max(0), "nq mean", b[:, 5].mean(), "total span", (b[:, 4].max() - b[:, 0].min()))
