"""dev tool: psx_match wall time for two 1080p-sized descriptor sets (18.5k x 18.5k)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from popsift_amd import capi
L = capi.lib()
L.psx_match.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
for n in (2000, 18500):
    a = torch.rand(n, 128, device="cuda"); b = torch.rand(n, 128, device="cuda"); torch.cuda.synchronize()
    mm = np.zeros((n, 3), np.int32); dd = np.zeros((n, 2), np.float32)
    for rep in range(3):
        t = time.perf_counter()
        rc = L.psx_match(0, C.c_void_p(a.data_ptr()), n, C.c_void_p(b.data_ptr()), n, mm.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p))
        dt = time.perf_counter() - t
    print("n=%d: %.2f ms  (%.1f G pairs/s)" % (n, dt * 1e3, n * n / dt / 1e9))
