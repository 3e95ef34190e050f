"""Measurement build only (tools/build_phase_lib.sh): k_descriptors with its LDS atomics and / or its gradient loads removed."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import capi
from popsift_amd.synth import synth
L = capi.lib()
img = synth(1920, 1080, 1000)
ctx = capi.Context(capi.default_config(octaves=5)); ctx.upload(img)
ctx.enable_timers(True)
for dbg in (0, 1, 2, 3):
    L.psx_debug_set_desc_dbg(dbg)
    ts = []
    for i in range(8):
        ctx.extract(); ts.append(ctx.stage_times()[3])
    print("dbg", dbg, "(1 = no atomics, 2 = no loads): descriptors stage %.1f us" % (sorted(ts)[len(ts)//2] * 1e3))
