"""dev tool (GPU box, phase-timing build: tools/build_phase_lib.sh, POPSIFT_HIP_LIB=popsift_amd/lib_phase/libpopsift_hip.so):
where the cycles of k_level0_x2 go -- clock64 stamps per workgroup around its five phases (the same stamps as k_blur's,
tools/blur_phase.py): commit (texel conversion, the 4 x 4 blocks of U into LDS) + the previous step's stores; wait at the first
barrier; issue of the next step's texel loads + H pass; wait at the second barrier; V pass."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from popsift_amd import capi
from popsift_amd.synth import synth
L = capi.lib()
ctx = capi.Context(capi.default_config(octaves=5, sift_mode=2)); ctx.upload(synth(1920, 1080, 1000)); ctx.extract(); ctx.sync()
buf = torch.zeros(1100 * 8, dtype=torch.int64, device="cuda")
L.psx_debug_set_level0_buffer(C.c_void_p(buf.data_ptr()))
ctx.enable_blur_probe(True)
rows = []
for i in range(6):
    buf.zero_(); torch.cuda.synchronize()
    ctx.extract(); ctx.sync()
    l0_ms = ctx.probe_extra_times()[0]
    b = buf.cpu().numpy().reshape(-1, 8); b = b[b[:, 6] > 0]
    if i >= 2: rows.append((l0_ms, b))
L.psx_debug_set_level0_buffer(None)
ms = np.mean([r[0] for r in rows]); b = np.concatenate([r[1] for r in rows])
tot = b[:, 5].mean()
print("k_level0_x2 (3840 x 2160, u8): %.2f us per launch (events around the launch), %d workgroups, %.1f steps each" % (ms * 1e3, len(rows[0][1]), b[:, 6].mean()))
print("mean cycles per workgroup: commit + stores %.0f (%.0f %%)  barrier 1 %.0f (%.0f %%)  loads + H %.0f (%.0f %%)  barrier 2 %.0f (%.0f %%)  V %.0f (%.0f %%)  | total %.0f (%.2f us at 2.4 GHz), prologue + epilogue %.0f" % (
    b[:, 0].mean(), 100 * b[:, 0].mean() / tot, b[:, 1].mean(), 100 * b[:, 1].mean() / tot, b[:, 2].mean(), 100 * b[:, 2].mean() / tot,
    b[:, 3].mean(), 100 * b[:, 3].mean() / tot, b[:, 4].mean(), 100 * b[:, 4].mean() / tot, tot, tot / 2400.0, tot - b[:, :5].sum(1).mean()))
q = np.percentile(b[:, 5], [5, 50, 95])
print("workgroup total cycles: p5 %.0f  p50 %.0f  p95 %.0f" % tuple(q))
