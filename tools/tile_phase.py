"""In-kernel clock64 stamps of k_blur_tile per phase (measurement build: tools/build_phase_lib.sh, then
POPSIFT_HIP_LIB=popsift_amd/lib_phase/libpopsift_hip.so POPSIFT_TILE=1 python tools/tile_phase.py).
Stamps per workgroup: 0 entry, 1 job header read, 2 loads issued + LDS stores done, 3 barrier, then per level
H done / barrier / V done.  Printed per launch shape (the LAST frame's launches overwrite each other per block index,
so run with the octaves of interest only: POPSIFT_TILE_MAXPX)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from popsift_amd import capi
from popsift_amd.synth import synth
L = capi.lib()
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img); ctx.extract(); ctx.sync()
buf = torch.zeros(16 + 4096 * 16, dtype=torch.int64, device="cuda")
L.psx_debug_set_tile_buffer(C.c_void_p(buf.data_ptr()))
for rep in range(3):
    buf.zero_(); torch.cuda.synchronize()
    ctx.extract(); ctx.sync()
    a = buf.cpu().numpy()
    n = int(a[0])
    b = a[16:16 + n * 16].reshape(-1, 16)
    print("frame", rep, "workgroups", n)
    meta = b[:, 15]
    for key in sorted(set(meta.tolist())):
        sel = meta == key
        nlev, grid, tx, ty = key & 255, (key >> 8) & 0xffffff, (key >> 32) & 255, (key >> 40) & 255
        bb = b[sel]
        d = np.diff(bb[:, :4 + 3 * nlev], axis=1)
        m = d.mean(0)
        tot = (bb[:, 3 + 3 * nlev] - bb[:, 0]).mean()
        span = (bb[:, 3 + 3 * nlev].max() - bb[:, 0].min())
        print("  grid %4d  %d levels  tile %dx%d  %4d wgs: hdr %5.0f load %5.0f bar %5.0f |" % (grid, nlev, tx, ty, sel.sum(), m[0], m[1], m[2]),
              " ".join("H %5.0f bar %5.0f V %5.0f%s" % (m[3 + 3 * l], m[4 + 3 * l], m[5 + 3 * l], " |") for l in range(nlev)),
              "per wg %6.0f  first-in..last-out %6.0f cycles" % (tot, span))
L.psx_debug_set_tile_buffer(None)
