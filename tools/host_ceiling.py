#!/usr/bin/env python3
"""What the HOST side of the C++ API sustains without the kernels -- the ceiling that bends the 1 -> 8 GPU curve,
measured on one GPU (task 4).

N processes, each one PopSift replica (its dispatcher thread + PIPE_DEPTH workers, pinned pools, queues) on device 0,
run the end-to-end loop of bench.py (host frame -> PopSift::enqueue -> SiftJob::get -> FeaturesHost) with
PSX_NULL_DEVICE_WORK (api.hip): after a context's first frame psx_extract launches nothing and the first frame's results
(15 k keypoints, 18.7 k descriptors of the 1080p bench frame) stand in for every later frame.
    mode 1: the uploads (2 MB), the counter read-back and the result downloads (10 MB) still run: host + PCIe of ONE link
    mode 2: no DMA either: threads, queues, pools, the image copy into the pinned job buffer, the per-keypoint record loop
Per configuration: frames/s of all processes, host CPU milliseconds per frame (user + system, all threads), pinned bytes
per replica.  8 GPUs at the 1-GPU rate R need 8 R frames/s from the host: mode 2 says whether the host SOFTWARE
delivers that on this box's cores, mode 1 what one PCIe link carries (on an 8-GPU node every GPU has its own).

    python tools/host_ceiling.py [--seconds S] [--procs 1,8] [--modes 1,2]         one JSON line
"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 1920, 1080
MAX_OUT = 24


def worker(mode, seconds, t_start):
    os.environ["PSX_NULL_DEVICE_WORK"] = str(mode)       # read at psx_create
    import numpy as np
    from popsift_amd import capi
    from popsift_amd.synth import synth
    base = synth(W, H, 1000)
    frames = [base, np.ascontiguousarray(base[:, ::-1]), np.ascontiguousarray(np.roll(base, 97, 1)), np.ascontiguousarray(base[::-1])]
    ps = capi.PopSift(capi.default_config(octaves=5, sift_mode=2), device=0)
    jobs = []
    for i in range(96):                                   # prime: every worker context sees real frames first
        jobs.append(ps.enqueue(frames[i % 4]))
        if len(jobs) >= MAX_OUT:
            ps.get_counts(jobs.pop(0))
    while jobs:
        ne, no = ps.get_counts(jobs.pop(0))
    while time.time() < t_start:
        time.sleep(0.005)
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            if len(jobs) >= MAX_OUT:
                ps.get_counts(jobs.pop(0)); n += 1
            jobs.append(ps.enqueue(frames[(n + len(jobs)) % 4]))
    while jobs:
        ps.get_counts(jobs.pop(0)); n += 1
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    st = capi.pool_stats(0)
    print(json.dumps({"frames": n, "seconds": dt, "cpu_s": (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime),
                      "keypoints": ne, "descriptors": no, "pinned_bytes": st["free_bytes"] + st["in_use"], "pool_allocs": st["allocs"]}), flush=True)
    ps.close()


def measure(seconds=1.5, procs=(1, 8), modes=(1, 2), timeout=120, configs=None):
    """configs: explicit list of (mode, processes) instead of the modes x procs product"""
    out = {"cores": len(os.sched_getaffinity(0)), "frame": "1920x1080 u8, config 1's Config (VLFeat mode)", "runs": []}
    for mode, n in (configs if configs is not None else [(m, q) for m in modes for q in procs]):
        if True:
            t_start = time.time() + 3.0 + 0.25 * n        # import + HIP init + priming of every process (late starters still overlap)
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(mode), str(seconds), repr(t_start)],
                                   cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(n)]
            res, err = [], None
            for p in ps:
                try:
                    so, se = p.communicate(timeout=timeout)
                    res.append(json.loads(so.strip().splitlines()[-1]))
                except Exception as e:                     # never lose the bench line to this leg
                    err = "%s: %s" % (type(e).__name__, str(e)[:200])
                    p.kill()
            if err or not res:
                out["runs"].append({"mode": mode, "processes": n, "failed": err}); continue
            fps = sum(r["frames"] / r["seconds"] for r in res)
            frames = sum(r["frames"] for r in res)
            cpu = sum(r["cpu_s"] for r in res)
            out["runs"].append({"mode": mode, "processes": n, "frames_per_s": round(fps, 1), "mpix_per_s": round(fps * W * H / 1e6, 1),
                                "host_cpu_ms_per_frame": round(cpu / frames * 1e3, 4), "host_cores_busy": round(cpu / max(r["seconds"] for r in res), 2),
                                "pinned_MB_per_replica": round(max(r["pinned_bytes"] for r in res) / 2 ** 20, 1),
                                "keypoints_per_frame": res[0]["keypoints"], "descriptors_per_frame": res[0]["descriptors"],
                                "result_MB_per_frame": round((res[0]["keypoints"] * 52 + res[0]["descriptors"] * 512) / 1e6, 2)})
    out["what"] = ("N replicas (processes) on ONE GPU with the kernels skipped (PSX_NULL_DEVICE_WORK): mode 1 keeps the upload / download "
                   "DMAs (one PCIe link shared by all processes), mode 2 skips them too (host software only)")
    return out


def main():
    a = sys.argv[1:]
    if a and a[0] == "worker":
        worker(int(a[1]), float(a[2]), float(a[3]))
        return
    seconds, procs, modes = 1.5, (1, 8), (1, 2)
    i = 0
    while i < len(a):
        if a[i] == "--seconds": seconds = float(a[i + 1]); i += 2
        elif a[i] == "--procs": procs = tuple(int(v) for v in a[i + 1].split(",")); i += 2
        elif a[i] == "--modes": modes = tuple(int(v) for v in a[i + 1].split(",")); i += 2
        else: i += 1
    print(json.dumps(measure(seconds, procs, modes)))


if __name__ == "__main__":
    main()
