#!/usr/bin/env python3
"""bench.py -- end-to-end SIFT extraction throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): 1920x1080 grayscale u8 frames, default Config with
octaves=5, levels=3 (x2 upsample => octave 0 is 3840x2160).  A "step" is one batch of
BATCH distinct synthetic frames, already resident in HBM, pushed through the full hot path
(pyramid -> extrema -> orientation -> descriptors).  Frames are independent, so ranks share
nothing: each rank processes its own BATCH frames per step (weak scaling, no collective on the
data path); value = total pixels of all ranks / max-over-ranks time.

Two timed legs of K steps each, same frames, same kernels:
  value        : inputs AND results resident in HBM (Feature records + descriptors stay in the
                 context's device buffers, the reference's FeaturesDev / MatchingMode end state;
                 only the two counters are read back per frame).  No PCIe inside the timed region.
  host_export  : as above, plus every frame's Feature records and descriptors delivered into pinned
                 host memory inside the timed region (psx_attach_export: the kernels stream them over
                 PCIe; the reference's FeaturesHost / ExtractingMode end state, Pyramid::get_descriptors).

The JSON line also carries
  roofline     : the separable-Gaussian kernel (k_blur, octave 0): algorithmic bytes (8 B/pixel)
                 / average launch duration measured with HIP events on the kernel's stream
  cpu_baseline : the CPU oracle (port of the reference arithmetic, OpenMP over all host cores)
                 timed on a bounded sample of the same frames (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
BATCH = 8          # frames per step per rank
NCTX = int(os.environ.get("POPSIFT_BENCH_CTX", "16"))   # extraction contexts (pyramids + streams) in flight per GPU
HBM_PEAK_GBS = 8000.0


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from popsift_amd import capi
    from popsift_amd.synth import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # synthetic frames, resident in HBM before the timed region
    frames_np = [synth(W, H, 1000 + rank * BATCH + i) for i in range(BATCH)]
    frames = [torch.from_numpy(f).to(dev) for f in frames_np]
    torch.cuda.synchronize()

    cfg = capi.default_config(octaves=5)
    ctxs = [capi.Context(cfg, device=local_rank) for _ in range(NCTX)]
    # pinned host destinations, one per context (Pyramid::get_descriptors downloads into pinned memory)
    cap_f, cap_d = 100000, 200000
    pin_f = [torch.empty(cap_f * capi.FEATURE_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(NCTX)]
    pin_d = [torch.empty(cap_d * 128, dtype=torch.float32).pin_memory() for _ in range(NCTX)]

    def set_export(on):
        # zero-copy export: the kernels stream features / descriptors into the pinned buffers over PCIe
        for c in range(NCTX):
            if on:
                ctxs[c].attach_export(pin_f[c], pin_d[c])
            else:
                ctxs[c].attach_export(None, None)

    def submit(c, i):
        ctxs[c].set_input_tensor(frames[i])
        ctxs[c].extract()

    def collect(c):
        # waits for the frame of context c; afterwards its results are in pin_f[c] / pin_d[c]
        return ctxs[c].counts()

    inflight = []          # contexts with a frame in flight, oldest first (persists across steps)
    state = {"next": 0}

    def step():
        """One step = BATCH frames submitted; results of older frames are collected as the ring of
        contexts fills up, so the pipeline stays full across step boundaries."""
        kp = 0
        for i in range(BATCH):
            if len(inflight) == NCTX:
                kp += collect(inflight.pop(0))[0]
            c = state["next"]
            state["next"] = (c + 1) % NCTX
            submit(c, i)
            inflight.append(c)
        return kp

    def drain():
        kp = 0
        while inflight:
            kp += collect(inflight.pop(0))[0]
        return kp

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(export):
        set_export(export)
        for _ in range(args.warmup):
            step()
        drain()
        barrier()
        t0 = time.perf_counter()
        kps = 0
        for _ in range(args.steps):
            kps += step()
        kps += drain()          # every frame of the K steps is complete (and collected) inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            kk = torch.tensor([kps], dtype=torch.float64, device=dev)
            dist.all_reduce(kk, op=dist.ReduceOp.SUM)
            kps = float(kk.item())
        return dt, float(kps)

    dt, kps_total = timed_leg(False)        # the headline leg: everything stays in HBM
    dt_x, _ = timed_leg(True)               # same work + delivery to pinned host memory
    set_export(False)

    n_frames = world * BATCH * args.steps
    mpix_s = n_frames * W * H / dt / 1e6

    if rank == 0:
        # ---- roofline of the dominant kernel: separable Gaussian, octave 0, levels 1..L-1 ----
        c0 = ctxs[0]
        c0.sync()
        tot_ms, tot_bytes, nl = 0.0, 0.0, 0
        for lvl in range(1, c0.num_levels):
            c0.time_blur(0, lvl, 5)                     # warm
            ms, by = c0.time_blur(0, lvl, 50)
            tot_ms += ms
            tot_bytes += by
            nl += 1
        achieved = tot_bytes / (tot_ms * 1e-3) / 1e9     # GB/s, algorithmic 8 B/pixel
        # what the memory system delivers on this box (torch is only the memcpy + event timer here):
        #   measured_copy       a 1 GiB -> 1 GiB device copy (the "measured HBM roofline"; the
        #                       working set is far beyond the 256 MB Infinity Cache)
        #   measured_copy_plane a copy of one octave-0 plane (33 MB), i.e. what "read a plane, write a
        #                       plane" costs with the plane sizes k_blur actually works on
        def copy_rate(nfloats, reps):
            src_t = torch.rand(nfloats, device=dev)
            dst_t = torch.empty_like(src_t)
            for _ in range(3):
                dst_t.copy_(src_t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                dst_t.copy_(src_t)
            e1.record()
            torch.cuda.synchronize()
            return 2.0 * nfloats * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        pw, ph = c0.octave_dims(0)
        copy_gbs = copy_rate(1 << 28, 10)
        copy_plane_gbs = copy_rate(pw * ph, 50)
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE
        # and --pmc WRITE_SIZE in separate runs, 2x FETCH_SIZE correction); null when not collected
        traffic = None
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))["k_blur_octave0_hbm_bytes_per_launch"]
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": "k_blur (octave 0, 3840x2160, levels 1..5)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "avg_launch_ms": round(tot_ms / nl, 5), "bytes_per_launch": tot_bytes / nl,
                    "traffic": traffic,
                    "measured_copy": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                    "measured_copy_plane": round(copy_plane_gbs, 1)}

        # one frame at a time on one context (BASELINE config 2, "single frame"): wall-clock latency
        c0.set_input_tensor(frames[0])
        lat = []
        for i in range(25):
            t1 = time.perf_counter()
            c0.extract()
            c0.counts()
            lat.append(time.perf_counter() - t1)
        lat = sorted(lat[5:])
        single_ms = lat[len(lat) // 2] * 1e3

        # per-stage device time of one frame (HIP events on the context's stream)
        c0.enable_timers(True)
        c0.set_input_tensor(frames[0])
        c0.extract()
        stages = c0.stage_times()
        c0.enable_timers(False)

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle as po
            ocfg = po.default_config(octaves=5)
            ncores = usable_cores()
            po.run(ocfg, frames_np[0], threads=ncores).close()     # warm
            n_s = 0
            t1 = time.perf_counter()
            while n_s < BATCH and (n_s < 2 or time.perf_counter() - t1 < 12.0):
                r = po.run(ocfg, frames_np[n_s], threads=ncores)
                r.close()
                n_s += 1
            cdt = time.perf_counter() - t1
            cpu = {"value": round(n_s * W * H / cdt / 1e6, 2), "unit": "Mpix/s", "cores": ncores,
                   "kind": "port", "sample": "%d frames 1920x1080, oracle/liboracle.so with OpenMP" % n_s}
            # OpenCV's CPU SIFT next to it when the box has cv2; this image does not ship it
            try:
                import cv2
                cv2.setNumThreads(ncores)
                sift = cv2.SIFT_create(nOctaveLayers=3, contrastThreshold=0.04, edgeThreshold=10, sigma=1.6)
                t2 = time.perf_counter()
                nk = len(sift.detectAndCompute(frames_np[0], None)[0])
                cpu["opencv_sift"] = {"value": round(W * H / (time.perf_counter() - t2) / 1e6, 2), "unit": "Mpix/s",
                                      "cores": ncores, "keypoints": nk, "sample": "1 frame 1920x1080"}
            except Exception:
                cpu["opencv_sift"] = "absent (cv2 is not installed in this image)"

        out = {
            "metric": "Mpixels/sec end-to-end SIFT (5 oct, 3 lvl/oct) + keypoints/sec",
            "value": round(mpix_s, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "1920x1080 u8 frames, default Config, octaves=5, levels=3, "
                                   "upscale x2 (octave 0 = 3840x2160), full pipe (pyramid, extrema, orientation, "
                                   "descriptors), inputs and results resident in HBM",
                       "frames_per_step_per_gpu": BATCH, "contexts_per_gpu": NCTX,
                       "parallelism": "replicas x%d (one image per GPU, no collective)" % world},
            "keypoints_per_s": round(kps_total / dt, 1),
            "keypoints_per_frame": round(kps_total / n_frames, 1),
            "ms_per_frame": round(dt / (BATCH * args.steps) * 1e3, 4),
            "single_frame": {"ms": round(single_ms, 4), "value": round(W * H / single_ms / 1e3, 1), "unit": "Mpix/s",
                             "what": "median wall time of one frame at a time on one context (no overlap between frames)"},
            "stage_ms_single_frame": {"pyramid": round(stages[0], 4), "extrema": round(stages[1], 4),
                                      "orientation": round(stages[2], 4), "descriptors": round(stages[3], 4)},
            "host_export": {"value": round(n_frames * W * H / dt_x / 1e6, 1), "unit": "Mpix/s",
                            "ms_per_step": round(dt_x / args.steps * 1e3, 4),
                            "what": "same steps with Feature records + descriptors streamed into pinned host memory"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)

    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
