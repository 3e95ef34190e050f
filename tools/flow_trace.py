#!/usr/bin/env python3
"""Per-job timeline of k_pyramid_flow from psx_flow_trace (GPU): for every (octave, level) the first dequeue, first start,
last end, and the mean wait / arithmetic / publish time of its items, in microseconds from the kernel's first dequeue.
  python tools/flow_trace.py [W H]      (environment: POPSIFT_FLOW_*)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from popsift_amd import capi
    from popsift_amd.synth import synth
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    ctx = capi.Context(capi.default_config(octaves=5, sift_mode=2))
    ctx.upload(synth(w, h, 1000))
    for _ in range(3):
        ctx.extract(); ctx.counts()
    L = capi.lib()
    L.psx_flow_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    n = C.c_int()
    L.psx_flow_trace(ctx._h, None, 0, C.byref(n))
    if n.value == 0:
        print("flow kernel not in use"); return
    buf = np.zeros((n.value, 6), np.int64)
    for rep in range(2):
        rc = L.psx_flow_trace(ctx._h, buf.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        assert rc == 0
    t = buf[:, :4].astype(np.float64) / 100.0        # us
    t0 = t[:, 0].min()
    t -= t0
    octv, lvl = (buf[:, 4] >> 40) & 0xff, (buf[:, 4] >> 32) & 0xff
    print("items %d, kernel span %.1f us, workgroups %d" % (n.value, t[:, 3].max(), len(np.unique(buf[:, 5]))))
    print(" o l  items | dequeue first..last | start first..last |  end first..last | wait mean/max | body mean/max | publish mean")
    for o in sorted(set(octv)):
        for l in sorted(set(lvl[octv == o])):
            m = (octv == o) & (lvl == l)
            a = t[m]
            print(" %d %d %6d | %7.1f %7.1f | %7.1f %7.1f | %7.1f %7.1f | %6.2f %6.2f | %6.2f %6.2f | %5.2f" % (
                o, l, m.sum(), a[:, 0].min(), a[:, 0].max(), a[:, 1].min(), a[:, 1].max(), a[:, 3].min(), a[:, 3].max(),
                (a[:, 1] - a[:, 0]).mean(), (a[:, 1] - a[:, 0]).max(), (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).max(),
                (a[:, 3] - a[:, 2]).mean()))
    # where do slow items come from?  octave 0, level 1: arithmetic time by strip position, by class, percentiles
    m = (octv == 0) & (lvl == 1)
    body = (t[:, 2] - t[:, 1])[m]
    strip = (buf[:, 4] & 0xffff)[m]
    cls = (buf[:, 5] & 7)[m]
    print("o0 L1 body percentiles 10/50/90/99/max: %s" % np.round(np.percentile(body, [10, 50, 90, 99, 100]), 2).tolist())
    print("  edge strips (first / last) mean %.2f, interior mean %.2f" % (body[(strip == 0) | (strip == strip.max())].mean(), body[(strip > 0) & (strip < strip.max())].mean()))
    print("  by class blockIdx & 7: %s" % [round(float(body[cls == q].mean()), 2) for q in range(8)])
    wg = buf[:, 5][m]
    first = np.array([t[m][wg == w_][:, 1].min() for w_ in np.unique(wg)])
    print("  workgroups that ran L1 items: %d; their first start %.2f .. %.2f us" % (len(first), first.min(), first.max()))
    busy = (t[:, 2] - t[:, 1]).sum()
    wait = (t[:, 1] - t[:, 0]).sum()
    print("sum of arithmetic %.0f wg-us, of waits %.0f wg-us, of publish %.0f wg-us; span x workgroups %.0f" % (
        busy, wait, (t[:, 3] - t[:, 2]).sum(), t[:, 3].max() * len(np.unique(buf[:, 5]))))
    ctx.close()


if __name__ == "__main__":
    main()
