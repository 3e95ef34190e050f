#!/usr/bin/env python3
"""bench.py -- end-to-end SIFT extraction throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1] / BASELINE.md workload 2): 1920x1080 grayscale u8 frames, config 1's Config
(default Config, setOctaves(5), levels=3, setMode(VLFeat): BASELINE.md section 3; x2 upsample => octave 0 is
3840x2160), distinct frames streamed through the product's public API: host image -> PopSift::enqueue ->
SiftJob::get -> FeaturesHost (the C++14 library popsift_amd/lib/libpopsift.so, bound through include/popsift_c.h).
A "step" is BATCH = 320 frames per GPU = five passes over the rank's 64 distinct frames (BASELINE config 4's batch
five times; the driver's 20 steps time 6400 frames ~ 2.3 s: SURVEY.md 8d asks for >= 200 frames).  Frames are
independent: frame i of the global sequence goes to GPU i mod N (BASELINE config 4), ranks share nothing, no
collective on the data path; default = weak scaling (BATCH frames per GPU and step); --strong = BASELINE config 4's
literal shape (BATCH frames per step in TOTAL, BATCH / N per GPU).  value = total pixels of all ranks /
max-over-ranks time, results of every timed frame collected inside the timed region.

Legs (same frames, same kernels, K steps each):
  value / end_to_end : host frames in, FeaturesHost out (upload + full pipe + results in host memory)
  device_resident    : C-ABI, inputs AND results resident in HBM (no PCIe in the timed region)
  host_export        : C-ABI, inputs resident, Feature records + descriptors streamed to pinned host memory
The JSON line also carries
  roofline     : the separable-Gaussian kernel (k_blur, octave 0, levels 1..5): algorithmic bytes (8 B/pixel)
                 / average duration of those launches measured IN the pipeline (begin/end events of each
                 dispatch inside real psx_extract calls), next to the measured HBM copy rate of a hand-written
                 16 B/lane copy kernel over 1 GiB
  config3      : BASELINE config 3 (4096x4096, 6 octaves, 8192x8192 octave 0), device resident, one context
  cpu_baseline : the CPU oracle (port of the reference arithmetic, OpenMP over the host cores) timed on a
                 bounded sample of the same frames (rank 0, N=1 only)
  parity_checked : 4 of the timed frames of EVERY rank re-run through the SAME PopSift object after the timed region and
                 matched against the oracle (exact mismatch counts, summed over ranks)
  popsift_mode : the end-to-end leg with setMode(PopSift) (the Config rounds 1-3 quoted as `value`)
  caller_profile : the end-to-end leg in the shape of the external caller SURVEY.md 8b names (AliceVision):
                 PopSift(config, ExtractingMode, FloatImages), float frames (4 bytes / pixel over PCIe),
                 setFilterMaxExtrema + LargestScaleFirst (one host counter read per frame), setNormalizationMultiplier(9),
                 with its own pcie_gbs and parity_checked
  sustained    : the end-to-end leg kept running for >= 3 s / >= 8000 frames (N=1): Mpix/s per 256-frame window
                 (min / median / max), GPU clock before and after
  sparse_frames: the end-to-end leg on a second frame set with ~2 keypoints / 1000 px (the default set has ~7)
  pcie_gbs, pipe_roofline, roofline.stage, alt_modes_ms: see DESIGN.md section 6
"""
import argparse
import json
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
BATCH = 320        # frames per step per rank: five passes over the NDISTINCT distinct frames (20 driver steps = 6400 timed frames, > 2 s)
NBASE = 16         # synthetic base frames per rank; the other distinct frames are shifted / flipped variants of them
NDISTINCT = 64     # distinct frames per rank that the steps cycle through
NCTX = int(os.environ.get("POPSIFT_BENCH_CTX", "16"))       # C-ABI legs: extraction contexts in flight per GPU
MAX_OUT = int(os.environ.get("POPSIFT_BENCH_OUTSTANDING", "24"))   # end-to-end leg: jobs outstanding per PopSift
HBM_PEAK_GBS = 8000.0
SUSTAINED_S = float(os.environ.get("POPSIFT_BENCH_SUSTAINED_S", "3.0"))
SUSTAINED_FRAMES = int(os.environ.get("POPSIFT_BENCH_SUSTAINED_FRAMES", "8000"))
WINDOW = 256                                                 # frames per sustained-leg window
SUB = 64                                                     # the sustained leg advances in sub-steps of SUB frames

# Config profiles of the end-to-end legs (keyword overrides of capi.default_config)
HEADLINE_KW = dict(octaves=5, sift_mode=2)                   # BASELINE config 1's Config: setOctaves(5), setMode(VLFeat)
POPSIFT_KW = dict(octaves=5)                                 # setMode(PopSift), the default: rounds 1-3 quoted this one
# the external caller of SURVEY.md 8b (AliceVision's popSIFT describer): FloatImages, grid filter with
# LargestScaleFirst, descriptors scaled by 2^9
CALLER_KW = dict(octaves=5, filter_max_extrema=10000, grid_filter_mode=1, norm_multi=9)
PROFILES = {"headline": (HEADLINE_KW, False), "popsift": (POPSIFT_KW, False), "caller": (CALLER_KW, True)}
SIFT_MODE_NAMES = {0: "PopSift", 1: "OpenCV", 2: "VLFeat"}


def caller_frames(frames_u8):
    """Float frames in [0, 1) as PopSift::FloatImages expects (popsift.h:68-73): v / 256 of the u8 frames."""
    import numpy as np
    return [(f.astype(np.float32) / np.float32(256.0)).astype(np.float32) for f in frames_u8]


def octave_pixels(w, h, octaves, up=1):
    """Pixels of every octave's planes for a w x h input (popsift.cpp:124-125, sift_pyramid.cu:132-133)."""
    import math
    W0, H0 = int(math.ceil(w * 2 ** up)), int(math.ceil(h * 2 ** up))
    out = []
    for _ in range(octaves):
        out.append(W0 * H0)
        W0, H0 = (W0 + 1) // 2, (H0 + 1) // 2
    return out


def algorithmic_bytes(w, h, octaves, levels=3, input_bytes_per_px=1):
    """SURVEY.md 8d.  stage: the separable-Gaussian stage alone (8 B per pixel and level: octave 0 has levels+2
    plane-to-plane blurs + level 0 from the input = 4 N written; octaves > 0 a decimation + levels+2 blurs);
    pipe: A_min of the whole image pipe (+ 24 B per pixel for the fused DoG / extrema scan)."""
    px = octave_pixels(w, h, octaves)
    L = levels + 3
    stage = (8 * (L - 1) + 4) * px[0] + (8 * (L - 1) + 8) * sum(px[1:]) + w * h * input_bytes_per_px
    pipe = stage + 24 * sum(px)
    return stage, pipe


def kernel_sources_sha1():
    """SHA-1 over the sources of the roofline kernel (k_blur): what profiles/pmc_summary.json must have been collected from."""
    import hashlib
    h = hashlib.sha1()
    for f in ("pyramid.hip", "blur_arith.h"):
        h.update(open(os.path.join(ROOT, "popsift_amd", "csrc", "hip", f), "rb").read())
    return h.hexdigest()


def read_sclk_mhz(device=0):
    """Current shader clock from sysfs: the line marked '*' in pp_dpm_sclk.  The card index of the visible device is
    not known inside the container (the box exposes every card's sysfs node), so this returns the HIGHEST current
    clock over all cards -- sampled while this process keeps its GPU busy, that is this GPU's.  None when unreadable."""
    import glob
    best = None
    for path in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(path).read().splitlines():
                if line.strip().endswith("*"):
                    mhz = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
                    best = mhz if best is None else max(best, mhz)
        except Exception:
            pass
    return best


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


def pin_process_to_gpu_numa_node(capi, device):
    """One process per GPU: run this rank (the caller thread that copies frames into pinned job buffers, the workers
    inherit it) on the CPUs local to its GPU's PCIe root, so that on a two-socket 8-GPU host no frame crosses the socket
    link on its way to the GPU.  Only on multi-node hosts, only when sysfs names the CPUs, never widening the affinity."""
    try:
        if not os.path.exists("/sys/devices/system/node/node1"):
            return
        import ctypes
        buf = ctypes.create_string_buffer(64)
        if capi.lib().psx_device_pci(device, buf, 64) != 0 or not buf.value:
            return
        path = "/sys/bus/pci/devices/%s/local_cpulist" % buf.value.decode().lower()
        cpus = set()
        for tok in open(path).read().strip().split(","):
            a, _, b = tok.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
    except Exception:
        pass


def frame_seed(j, rank, world):
    """Seed of this rank's j-th base frame: global frame i = j * world + rank goes to GPU i mod world."""
    return 1000 + j * world + rank


def make_frames(rank, world, synth, sparse=False):
    """NDISTINCT distinct frames for this rank: NBASE synthetic base frames (popsift_amd/synth.py; base 0 of
    rank 0 is the frame the parity tests check) and cheap distinct variants of them (cyclic shift + flip).
    sparse: the ~2 keypoints / 1000 px variant of the generator."""
    import numpy as np
    base = [synth(W, H, frame_seed(j, rank, world), sparse=True) if sparse else synth(W, H, frame_seed(j, rank, world))
            for j in range(NBASE)]
    frames = []
    for v in range(NDISTINCT // NBASE):
        for b in base:
            f = b if v == 0 else np.roll(b, (53 * v, 97 * v), axis=(0, 1))
            if v & 1:
                f = f[:, ::-1]
            frames.append(np.ascontiguousarray(f))
    return frames


class GpuBackend:
    """Everything of the bench that touches the GPU (tests/test_distributed_cpu.py swaps in a stub with the same
    interface to run the rank / seed / step / reduce control flow under gloo on CPU)."""
    dist_backend = "nccl"
    device_override = None      # --ranks-on-device D: every rank uses GPU D (the multi-process path on a one-GPU box)

    def __init__(self, rank, local_rank, world):
        import numpy as np
        import torch
        from popsift_amd import capi
        from popsift_amd.synth import synth
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
        if self.device_override is not None:
            local_rank = self.device_override
        torch.cuda.set_device(local_rank)
        pin_process_to_gpu_numa_node(capi, local_rank)
        self.np, self.torch, self.capi = np, torch, capi
        self.device = local_rank
        self.dev = torch.device("cuda", local_rank)
        self.frames_dense = make_frames(rank, world, synth)
        self.frames_sparse = make_frames(rank, world, synth, sparse=True)
        self.frames_float = None
        self.frames_np = self.frames_dense
        self.cfg = capi.default_config(**HEADLINE_KW)
        self.profile = "headline"
        self.ps = None
        self.ctxs = []
        self.desc_total = 0

    def sync(self):
        self.torch.cuda.synchronize()

    def reduce_tensor(self, v):
        # gloo (--dist-backend gloo) reduces host tensors; RCCL device tensors
        return self.torch.tensor([v], dtype=self.torch.float64, device="cpu" if self.dist_backend == "gloo" else self.dev)

    # ---- leg 1: the public C++ API, host frames in, FeaturesHost out ----
    def e2e_open(self, profile="headline"):
        kw, is_float = PROFILES[profile]
        self.profile = profile
        if is_float and self.frames_float is None:
            self.frames_float = caller_frames(self.frames_dense)
        self.frames_np = self.frames_float if is_float else self.frames_dense
        self.ps = self.capi.PopSift(self.capi.default_config(**kw), device=self.device, float_images=is_float)

    def e2e_enqueue(self, i):
        return self.ps.enqueue(self.frames_np[i % NDISTINCT])      # PopSift::enqueue (deep copy of the image)

    def e2e_get(self, job):
        ne, no = self.ps.get_counts(job)                           # SiftJob::get (blocks); FeaturesHost deleted
        self.desc_total += no
        return ne

    def e2e_select(self, which):
        """frame set of the following end-to-end legs: 'dense' (default, ~7 keypoints / 1000 px) or 'sparse' (~2)"""
        self.frames_np = self.frames_sparse if which == "sparse" else self.frames_dense

    def e2e_parity(self, indices):
        """Outside the timed region: the given timed frames once more through the SAME PopSift object, full results
        matched against the oracle (the checker; nothing here is timed or shipped)."""
        from oracle import pyoracle as po
        from tests.parity import match_features
        kw = PROFILES[self.profile][0]
        ocfg = po.default_config(**kw)
        scale = float(2 ** kw.get("norm_multi", 0))
        tot = {"frames": 0, "keypoints": 0, "descriptors": 0, "kp_miss": 0, "ori_miss": 0, "desc_miss": 0, "max_desc_dist": 0.0}
        jobs = [(i, self.ps.enqueue(self.frames_np[i % NDISTINCT])) for i in indices]
        for i, job in jobs:
            fb, db = self.ps.get(job)
            ref = po.run(ocfg, self.frames_np[i % NDISTINCT])
            fa, da = ref.features(), ref.descriptors()
            m = match_features(fa, da, fb, db, norm_scale=scale)
            tot["frames"] += 1; tot["keypoints"] += len(fa); tot["descriptors"] += len(da)
            tot["kp_miss"] += m["kp_miss"] + abs(len(fa) - len(fb)); tot["ori_miss"] += m["ori_miss"]; tot["desc_miss"] += m["desc_miss"]
            tot["max_desc_dist"] = round(max(tot["max_desc_dist"], m["max_desc_dist"]), 6)
            ref.close()
        tot["frame_indices"] = list(indices)
        return tot

    def e2e_close(self):
        self.ps.close()
        self.ps = None

    # ---- legs 2, 3: C-ABI, inputs already resident in HBM ----
    def abi_open(self):
        torch, capi = self.torch, self.capi
        self.frames = [torch.from_numpy(f).to(self.dev) for f in self.frames_dense[:NDISTINCT]]
        torch.cuda.synchronize()
        self.ctxs = [capi.Context(self.cfg, device=self.device) for _ in range(NCTX)]
        cap_f, cap_d = 100000, 200000
        self.pin_f = [torch.empty(cap_f * capi.FEATURE_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(NCTX)]
        self.pin_d = [torch.empty(cap_d * 128, dtype=torch.float32).pin_memory() for _ in range(NCTX)]

    def abi_export(self, on):
        for c in range(NCTX):
            if on:
                self.ctxs[c].attach_export(self.pin_f[c], self.pin_d[c])
            else:
                self.ctxs[c].attach_export(None, None)

    def abi_submit(self, c, i):
        self.ctxs[c].set_input_tensor(self.frames[i])
        self.ctxs[c].extract()

    def abi_collect(self, c):
        return self.ctxs[c].counts()[0]

    def abi_close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def extras(self, args, world):
        return extras(args, self.capi, self.torch, self.np, self.ctxs, self.frames, self.frames_np, world, self.device)


def run(args, backend_cls=GpuBackend, out=sys.stdout):
    """The bench proper: rank / world from the environment, one process per GPU, three timed legs, rank 0
    prints ONE JSON line.  Returns the dict (rank 0) or None."""
    t_wall = {"start": time.perf_counter()}
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if getattr(args, "dist_backend", None):
        backend_cls.dist_backend = args.dist_backend
    if getattr(args, "ranks_on_device", None) is not None:
        backend_cls.device_override = args.ranks_on_device
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend_cls.dist_backend, rank=rank, world_size=world)
    else:
        dist = None
    be = backend_cls(rank, local_rank, world)
    t_wall["setup"] = time.perf_counter()

    def barrier():
        if dist is not None:
            if backend_cls.dist_backend == "nccl":
                dist.barrier(device_ids=[be.device if hasattr(be, "device") else local_rank])      # RCCL: name the rank's device explicitly
            else:
                dist.barrier()
        be.sync()

    def reduce_max_sum(dt, kps):
        if dist is None:
            return dt, float(kps)
        tt = be.reduce_tensor(dt)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        kk = be.reduce_tensor(kps)
        dist.all_reduce(kk, op=dist.ReduceOp.SUM)
        return float(tt.item()), float(kk.item())

    def timed(step, drain, probe=None):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; every frame of the K steps
        is complete (and collected) inside the timed region.  probe(): a counter sampled at both ends of the timed
        region (the warm-up is excluded); its difference is returned as the third value."""
        for _ in range(args.warmup):
            step()
        drain()
        barrier()
        p0 = probe() if probe else 0
        t0 = time.perf_counter()
        kps = 0
        for _ in range(args.steps):
            kps += step()
        kps += drain()
        barrier()
        dt, kk = reduce_max_sum(time.perf_counter() - t0, kps)
        return dt, kk, (probe() - p0 if probe else 0)

    def reduce_sum_dict(d, keys):
        """sum the given integer fields over ranks (parity counts of every rank)"""
        if dist is None:
            return d
        for k in keys:
            t = be.reduce_tensor(float(d[k]))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            d[k] = int(round(float(t.item())))
        t = be.reduce_tensor(float(d["max_desc_dist"]))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        d["max_desc_dist"] = round(float(t.item()), 6)
        return d

    # frames per step of THIS rank: weak scaling = BATCH each; --strong = BASELINE config 4's literal shape, BATCH
    # frames per step in total (frame i -> GPU i mod N)
    per_rank = BATCH if not args.strong else max(1, BATCH // world)
    jobs = deque()
    e2e = {"i": 0}

    def e2e_step(n=None):
        kp = 0
        for _ in range(per_rank if n is None else n):
            if len(jobs) >= MAX_OUT:
                kp += be.e2e_get(jobs.popleft())
            jobs.append(be.e2e_enqueue(e2e["i"]))
            e2e["i"] += 1
        return kp

    def e2e_drain():
        kp = 0
        while jobs:
            kp += be.e2e_get(jobs.popleft())
        return kp

    desc_probe = (lambda: be.desc_total) if hasattr(be, "desc_total") else None
    parity_idx = [0, NDISTINCT // 4 + 1, NDISTINCT // 2 + 2, 3 * NDISTINCT // 4 + 3]

    def parity_all_ranks():
        """every rank re-runs 4 of ITS timed frames through its own PopSift object and checks them against the oracle;
        the counts are summed over ranks"""
        from tests.parity import budget
        d = be.e2e_parity(parity_idx)
        d = reduce_sum_dict(d, ["frames", "keypoints", "descriptors", "kp_miss", "ori_miss", "desc_miss"])
        b = budget(d["keypoints"])
        d["within_budget"] = bool(d["kp_miss"] <= b["kp"] and d["ori_miss"] <= b["ori"] and d["desc_miss"] <= b["desc"])
        d["ranks_checked"] = world
        d["what"] = ("4 timed frames of every rank re-run through the same PopSift object after the timed region, matched "
                     "against oracle/ (tolerances 1e-3, budget 0 keypoints, 8 orientations / descriptors per 100 000 keypoints rounded down: tests/parity.py::budget)")
        return d

    # ---- leg 1 (headline): the public C++ API, host frames in, FeaturesHost out; config 1's Config (VLFeat mode) ----
    be.e2e_open("headline")
    dt_e2e, kps_e2e, descs_e2e = timed(e2e_step, e2e_drain, desc_probe)

    # ---- outside the timed region, same PopSift object: parity of timed frames, sustained run, sparse frame set ----
    parity = sustained = sparse = popsift_leg = caller_leg = None
    extra_legs = hasattr(be, "e2e_parity") and not args.no_extras
    quick = bool(getattr(args, "quick", False))      # the headline leg, its parity check and the C-ABI legs only
    if extra_legs:
        if not args.no_parity:
            parity = parity_all_ranks()
        if world == 1 and SUSTAINED_S > 0 and not quick:
            wins, clk, n_s, kp_s = [], [], 0, 0
            t_s0 = tw = time.perf_counter()
            while n_s < SUSTAINED_FRAMES or time.perf_counter() - t_s0 < SUSTAINED_S:
                kp_s += e2e_step(SUB)
                n_s += SUB
                if n_s % WINDOW == 0:
                    now = time.perf_counter()
                    wins.append(WINDOW * W * H / (now - tw) / 1e6)
                    tw = now
                    if len(wins) % 4 == 1:                    # shader clock WHILE the pipe is full (an idle GPU reads ~150 MHz)
                        c = read_sclk_mhz(local_rank)
                        if c:
                            clk.append(c)
            kp_s += e2e_drain()
            dt_s = time.perf_counter() - t_s0
            first, last = wins[:len(wins) // 4 or 1], wins[-(len(wins) // 4 or 1):]
            wins_sorted = sorted(wins)
            sustained = {"seconds": round(dt_s, 3), "frames": n_s, "value": round(n_s * W * H / dt_s / 1e6, 1), "unit": "Mpix/s",
                         "window_frames": WINDOW, "window_min": round(wins_sorted[0], 1),
                         "window_median": round(wins_sorted[len(wins) // 2], 1), "window_max": round(wins_sorted[-1], 1),
                         "first_quarter_mean": round(sum(first) / len(first), 1), "last_quarter_mean": round(sum(last) / len(last), 1),
                         "sclk_mhz_first": clk[0] if clk else None, "sclk_mhz_last": clk[-1] if clk else None,
                         "sclk_mhz_min": min(clk) if clk else None, "sclk_mhz_max": max(clk) if clk else None,
                         "keypoints_per_s": round(kp_s / dt_s, 1)}
        n_leg = world * per_rank * args.steps
        if not quick:
            be.e2e_select("sparse")
            e2e["i"] = 0
            dt_sp, kps_sp, _ = timed(e2e_step, e2e_drain)
            be.e2e_select("dense")
            sparse = {"value": round(n_leg * W * H / dt_sp / 1e6, 1), "unit": "Mpix/s",
                      "keypoints_per_frame": round(kps_sp / n_leg, 1),
                      "keypoints_per_1000px": round(kps_sp / n_leg / (W * H / 1000.0), 2),
                      "keypoints_per_s": round(kps_sp / dt_sp, 1),
                      "what": "the end-to-end leg on frames with ~2 keypoints / 1000 px (popsift_amd/synth.py sparse=True)"}
    be.e2e_close()

    if extra_legs and not quick:
        # ---- the same leg with setMode(PopSift): the Config rounds 1-3 quoted as `value` ----
        be.e2e_open("popsift")
        e2e["i"] = 0
        dt_pm, kps_pm, _ = timed(e2e_step, e2e_drain)
        be.e2e_close()
        popsift_leg = {"value": round(n_leg * W * H / dt_pm / 1e6, 1), "unit": "Mpix/s",
                       "keypoints_per_frame": round(kps_pm / n_leg, 1), "keypoints_per_s": round(kps_pm / dt_pm, 1),
                       "what": "the end-to-end leg with setMode(PopSift) (rounds 1-3 quoted this Config as `value`)"}
        # ---- the external caller's profile: FloatImages + grid filter (LargestScaleFirst) + 2^9 descriptors ----
        be.e2e_open("caller")
        e2e["i"] = 0
        dt_cp, kps_cp, descs_cp = timed(e2e_step, e2e_drain, desc_probe)
        caller_leg = {"value": round(n_leg * W * H / dt_cp / 1e6, 1), "unit": "Mpix/s",
                      "config": "PopSift(config, ExtractingMode, FloatImages), octaves=5, setFilterMaxExtrema(%d), "
                                "setFilterSorting(LargestScaleFirst), setNormalizationMultiplier(%d), float32 frames in [0,1)"
                                % (CALLER_KW["filter_max_extrema"], CALLER_KW["norm_multi"]),
                      "jobs_outstanding_per_gpu": MAX_OUT,
                      "keypoints_per_frame": round(kps_cp / n_leg, 1), "keypoints_per_s": round(kps_cp / dt_cp, 1),
                      "pcie_gbs": {"h2d": round(n_leg * W * H * 4 / dt_cp / 1e9, 2),
                                   "d2h": round((kps_cp * 52 + descs_cp * 512 * (world if world > 1 else 1)) / dt_cp / 1e9, 2)},
                      "parity_checked": None if args.no_parity else parity_all_ranks(),
                      "what": "the end-to-end leg as the external caller of SURVEY.md 8b drives it: 4 bytes / pixel over PCIe, "
                              "the grid filter's host-side counter read once per frame (stalls one worker, not the pipe)"}
        be.e2e_close()

    t_wall["e2e_legs"] = time.perf_counter()
    # ---- legs 2 and 3: C-ABI, inputs already resident in HBM ----
    be.abi_open()
    inflight = deque()
    state = {"next": 0}

    def abi_step():
        kp = 0
        for i in range(per_rank):
            if len(inflight) == NCTX:
                kp += be.abi_collect(inflight.popleft())
            c = state["next"]
            state["next"] = (c + 1) % NCTX
            be.abi_submit(c, i % NDISTINCT)
            inflight.append(c)
        return kp

    def abi_drain():
        kp = 0
        while inflight:
            kp += be.abi_collect(inflight.popleft())
        return kp

    be.abi_export(False)
    dt_dev, kps_dev, _ = timed(abi_step, abi_drain)
    be.abi_export(True)
    dt_x, _, _ = timed(abi_step, abi_drain)
    be.abi_export(False)

    n_frames = world * per_rank * args.steps
    rate = lambda dt: round(n_frames * W * H / dt / 1e6, 1)

    result = None
    if rank == 0:
        mode_name = SIFT_MODE_NAMES[HEADLINE_KW.get("sift_mode", 0)]
        result = {
            "metric": "Mpixels/sec end-to-end SIFT (5 oct, 3 lvl/oct) + keypoints/sec",
            "value": rate(dt_e2e), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_e2e / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: 1920x1080 u8 host frames (%d distinct per GPU) -> PopSift::enqueue -> SiftJob::get -> "
                                   "FeaturesHost (C++ API of libpopsift.so, upload and result delivery inside the timed "
                                   "region); config 1's Config: setOctaves(5), levels=3, setMode(%s), ByteImages, grid filter off, "
                                   "RootSift x1, upscale x2 (octave 0 = 3840x2160), full pipe (pyramid, extrema, orientation, "
                                   "descriptors)" % (NDISTINCT, mode_name),
                       "sift_mode": mode_name, "input": "u8 (ByteImages)", "grid_filter": "off",
                       "frames_per_step_per_gpu": per_rank, "frames_timed": n_frames,
                       "jobs_outstanding_per_gpu": MAX_OUT,
                       "pipe_depth": os.environ.get("POPSIFT_PIPE_DEPTH", "8 (4 when this replica's share of the host is below two cores)"),
                       "parallelism": "replicas x%d (frame i -> GPU i mod N, no collective)%s" % (
                           world, "; --strong: %d frames per step in total (BASELINE config 4's shape)" % (per_rank * world) if args.strong else "")},
            "keypoints_per_s": round(kps_e2e / dt_e2e, 1),
            "keypoints_per_frame": round(kps_e2e / n_frames, 1),
            "ms_per_frame": round(dt_e2e / (per_rank * args.steps) * 1e3, 4),
            "keypoints_per_1000px": round(kps_e2e / n_frames / (W * H / 1000.0), 2),
            "parity_checked": parity, "sustained": sustained, "sparse_frames": sparse,
            "popsift_mode": popsift_leg, "caller_profile": caller_leg,
            "device_resident": {"value": rate(dt_dev), "unit": "Mpix/s", "ms_per_step": round(dt_dev / args.steps * 1e3, 4),
                                "keypoints_per_s": round(kps_dev / dt_dev, 1), "contexts_per_gpu": NCTX,
                                "what": "C-ABI psx_extract, inputs and results resident in HBM (no PCIe in the timed region)"},
            "host_export": {"value": rate(dt_x), "unit": "Mpix/s", "ms_per_step": round(dt_x / args.steps * 1e3, 4),
                            "what": "C-ABI, inputs resident in HBM, Feature records + descriptors streamed into pinned host memory"},
        }
        if hasattr(be, "desc_total"):
            # PCIe inside the timed region of `value` (warm-up excluded): the u8 frame up, 52-byte records + 512-byte
            # descriptors down
            h2d = n_frames * W * H
            d2h = kps_e2e * 52 + descs_e2e * 512 * (world if world > 1 else 1)
            result["pcie_gbs"] = {"h2d": round(h2d / dt_e2e / 1e9, 2), "d2h": round(d2h / dt_e2e / 1e9, 2),
                                  "what": "host<->device bytes of the end-to-end leg's timed steps / its time (all GPUs; D2H from rank 0's descriptor count)"}
        stage_b, pipe_b = algorithmic_bytes(W, H, 5)
        result["pipe_roofline"] = {
            "algorithmic_bytes_per_frame": pipe_b, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "end_to_end": round(pipe_b * n_frames / dt_e2e / 1e9, 1), "device_resident": round(pipe_b * n_frames / dt_dev / 1e9, 1),
            "frac_end_to_end": round(pipe_b * n_frames / dt_e2e / 1e9 / HBM_PEAK_GBS / world, 4),
            "frac_device_resident": round(pipe_b * n_frames / dt_dev / 1e9 / HBM_PEAK_GBS / world, 4),
            "what": "A_min of SURVEY.md 8d (68 N0 + 72 sum N_o + input) x frames / time, per GPU fraction of 8 TB/s"}
        t_wall["abi_legs"] = time.perf_counter()
        if not args.no_extras and not quick:
            result.update(be.extras(args, world))
        t_wall["extras"] = time.perf_counter()
        # where this run's wall time went (the driver allows the default run about a minute)
        result["wall_s"] = {"setup_frames_and_import": round(t_wall["setup"] - t_wall["start"], 1),
                            "end_to_end_legs": round(t_wall["e2e_legs"] - t_wall["setup"], 1),
                            "c_abi_legs": round(t_wall["abi_legs"] - t_wall["e2e_legs"], 1),
                            "extras": round(t_wall["extras"] - t_wall["abi_legs"], 1),
                            "total": round(t_wall["extras"] - t_wall["start"], 1)}
        print(json.dumps(result), file=out, flush=True)

    be.abi_close()
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    return result


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE config 4's literal shape: BATCH frames per step in TOTAL (BATCH / N per GPU) instead of per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of 4 timed frames")
    ap.add_argument("--no-extras", action="store_true", help="only the three timed legs")
    ap.add_argument("--no-host-ceiling", action="store_true", help="skip the host_ceiling leg (it starts 9 helper processes)")
    ap.add_argument("--quick", action="store_true",
                    help="only the headline leg, the parity check of its frames and the two C-ABI legs (tests of the multi-process path)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default=None,
                    help="torch.distributed backend of the N > 1 launch (default nccl = RCCL); gloo reduces the two timing scalars on the host")
    ap.add_argument("--ranks-on-device", type=int, default=None,
                    help="every rank uses this GPU instead of LOCAL_RANK: runs the real multi-process path (GpuBackend, barriers, NUMA "
                         "pinning, per-rank parity) on a box with one GPU; the aggregate is then ONE GPU's rate shared by the ranks")
    ap.add_argument("--extras", action="store_true",
                    help="also the slow informational legs (every alternative Gauss / descriptor mode, BASELINE config 3, the full host-ceiling sweep)")
    return ap.parse_args(argv)


def main():
    run(parse_args())


# homography of the BASELINE config 5 stand-in (tests/test_gpu_configs.py::test_config5_warped_pair_opencv_mode uses the same pair)
_H5 = [[0.891, -0.125, 60.0], [0.125, 0.891, -20.0], [4.0e-5, -2.0e-5, 1.0]]


def config5_leg(capi, np, device):
    """BASELINE config 5 (Oxford boat / graffiti, OpenCV-mode Config; testScripts/testOxfordDataset.sh.in:48 is the
    reference's protocol).  The dataset cannot be fetched here (no network; it is not in the reference tree either), so
    the leg runs the protocol on a stand-in pair: a synthetic 800x600 frame and its copy under a known homography,
    setMode(OpenCV) + setGaussMode(opencv).  Reported as SURVEY.md 8d asks: descriptor parity of both images against
    the oracle (the checker, outside any timed region), repeatability of the HIP keypoints under the homography next to
    the oracle's, and the MatchingMode matcher's inlier rate."""
    try:
        from oracle import pyoracle as po
        from popsift_amd.synth import synth, warp_homography
        from tests.parity import match_features, repeatability
        w, h = 800, 600
        Hm = np.array(_H5)
        a = synth(w, h, 515)
        b = warp_homography(a, Hm)
        kw = dict(octaves=5, sift_mode=1, gauss_mode=3)
        out, par = [], {"keypoints": 0, "descriptors": 0, "kp_miss": 0, "ori_miss": 0, "desc_miss": 0, "max_desc_dist": 0.0}
        ms = []
        for img in (a, b):
            ctx = capi.Context(capi.default_config(**kw), device=device)
            ctx.upload(img)
            ctx.extract(); ctx.counts()
            t1 = time.perf_counter()
            for _ in range(5):
                ctx.extract(); ctx.counts()
            ms.append((time.perf_counter() - t1) / 5 * 1e3)
            fb, db = ctx.download()
            ctx.close()
            ref = po.run(po.default_config(**kw), img)
            fa, da = ref.features(), ref.descriptors()
            m = match_features(fa, da, fb, db)
            par["keypoints"] += len(fa); par["descriptors"] += len(da)
            par["kp_miss"] += m["kp_miss"] + abs(len(fa) - len(fb)); par["ori_miss"] += m["ori_miss"]; par["desc_miss"] += m["desc_miss"]
            par["max_desc_dist"] = round(max(par["max_desc_dist"], m["max_desc_dist"]), 6)
            out.append((fa, fb, db))
            ref.close()
        (ra, fa, da), (rb, fb, db) = out
        rep_hip, n_in = repeatability(fa, fb, Hm, w, h)
        rep_ref, _ = repeatability(ra, rb, Hm, w, h)
        xa = np.zeros((len(da), 2)); xb = np.zeros((len(db), 2))
        for f, x in ((fa, xa), (fb, xb)):
            for k in range(4):
                sel = f["num_ori"] > k
                x[f["desc_idx"][sel, k]] = np.stack([f["xpos"][sel], f["ypos"][sel]], 1)
        mm, _ = capi.match(da, db, device=device)
        acc = mm[:, 2] == 1
        p = np.concatenate([xa[acc], np.ones((int(acc.sum()), 1))], 1) @ Hm.T
        err = np.linalg.norm(p[:, :2] / p[:, 2:3] - xb[mm[acc, 0]], axis=1)
        return {"images": "stand-in for the Oxford pair (not obtainable here): synth(800, 600, seed 515) and its warp under a known homography",
                "config": "setMode(OpenCV), setGaussMode(opencv), octaves=5", "parity_vs_oracle": par,
                "repeatability_hip": round(rep_hip, 4), "repeatability_oracle": round(rep_ref, 4), "keypoints_inside": n_in,
                "matcher_accepted": int(acc.sum()), "matcher_inliers_2px": round(float((err < 2.0).mean()), 4),
                "ms_per_image": [round(v, 4) for v in ms]}
    except Exception as e:                                   # never lose the headline to an extra
        return "failed: %s" % e


def match_leg(capi, np, device):
    """MatchingMode's 2-NN matcher (match.hip) on two device-resident descriptor sets of the bench frame's size (what
    FeaturesDev::match sees): G pairs / s of psx_match (prefilter with f16 MFMA + exact evaluation of the survivors with the
    reference's operation tree; indices, flags and distances bit-identical to the reference's scan).  Beside it, for scale: the
    exact scan of every pair costs 165 lane instructions per pair, i.e. the VALU issue peak (256 CUs x 4 SIMDs x 16 lanes per clock x
    2.4 GHz) allows 238 G pairs / s."""
    try:
        rng = np.random.default_rng(5)
        n = 18432
        v = rng.random((2 * n, 128), dtype=np.float32) ** 4
        v = np.sqrt(v / v.sum(1, keepdims=True)).astype(np.float32)           # RootSift-like: non-negative, unit L2 norm
        l, r = capi.DeviceDescriptors(v[:n], device), capi.DeviceDescriptors(v[n:], device)
        for _ in range(5):                                   # clocks and the thread's scratch buffers settle
            l.match(r)
        ts = []
        for _ in range(30):
            t1 = time.perf_counter()
            l.match(r)
            ts.append(time.perf_counter() - t1)
        dt = sorted(ts)[len(ts) // 2]
        l.close(); r.close()
        gp = n * n / dt / 1e9
        return {"left": n, "right": n, "seconds": round(dt, 6), "gpairs_per_s": round(gp, 1), "calls_timed": len(ts), "statistic": "median",
                "exact_scan_valu_peak_gpairs_per_s": round(256 * 4 * 16 * 2.4e9 / 165 / 1e9, 1),
                "what": "psx_match on device-resident descriptors (host call to results in host memory): k_match_norms / k_match_cvt of both sides, seeding pass, "
                        "k_match_mfma (v_mfma_f32_32x32x16_f16 + proven error margin), k_match_exact on the candidates; bit-identical to "
                        "the reference's scan (tests/test_gpu_parity.py::test_match_bit_exact, ::test_match_mfma_prefilter_equals_exact_scan); "
                        "POPSIFT_MATCH_MFMA=0 = the exact scan of every pair (rounds 1-4: 63-82 G pairs/s)"}
    except Exception as e:
        return "failed: %s" % e


def extras(args, capi, torch, np, ctxs, frames, frames_np, world, device):
    """Rank-0 measurements outside the timed legs: roofline of the dominant kernel, single-frame latency and
    stage times, BASELINE config 3, CPU baseline."""
    ex = {}
    laps, t_lap = {}, [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        laps[name] = round(now - t_lap[0], 1)
        t_lap[0] = now
    c0 = ctxs[0]
    c0.sync()

    # ---- roofline of the dominant kernel, measured in the pipeline ----
    c0.enable_blur_probe(True)
    c0.set_input_tensor(frames[0])
    per_level = None
    nrep = 0
    l0_ms = x0_ms = 0.0
    for it in range(24):
        c0.extract()
        ms, by = c0.blur_probe_times()
        if it >= 4:                                         # first frames warm caches / clocks
            per_level = ms if per_level is None else [a + b for a, b in zip(per_level, ms)]
            a_ms, l0_by, b_ms, x0_by = c0.probe_extra_times()
            l0_ms += a_ms; x0_ms += b_ms
            nrep += 1
    c0.enable_blur_probe(False)
    l0_ms /= nrep; x0_ms /= nrep
    per_level = [m / nrep for m in per_level]
    avg_ms = sum(per_level) / len(per_level)
    achieved = by / (avg_ms * 1e-3) / 1e9                   # GB/s, algorithmic 8 B/pixel
    # isolated replay of each level (round-1 method), for comparison only
    iso = []
    for lvl in range(1, c0.num_levels):
        c0.time_blur(0, lvl, 3)
        iso.append(c0.time_blur(0, lvl, 30)[0])
    copy_gbs, copy_ms = capi.copy_bench(device, 1 << 30, 10)
    # roofline.traffic comes from a committed PMC pass (tools/collect_profiles.sh -> profiles/pmc_summary.json), which also
    # records the SHA-1 of the kernel's sources at collection time: when they have changed since, the number is STALE and
    # is not reported (null + traffic_stale) -- re-run tools/collect_profiles.sh in the commit that changes the kernel
    traffic, traffic_stale, traffic_src = None, None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        traffic_src = pm.get("source")
        traffic_stale = pm.get("kernel_sources_sha1") != kernel_sources_sha1()
        if not traffic_stale:
            traffic = pm["k_blur_octave0_hbm_bytes_per_launch"]
    except Exception:
        pass
    ex["roofline"] = {
        "bound": "hbm", "kernel": "k_blur (octave 0, 3840x2160, levels 1..5), timed inside psx_extract",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "avg_launch_ms": round(avg_ms, 5), "per_level_ms": [round(m, 5) for m in per_level],
        "timing_method": "begin / end timestamps of each k_blur dispatch (hipExtLaunchKernelGGL events on the context's stream) inside real "
                         "psx_extract calls, 20 frames averaged; rocprofv3 --kernel-trace of the same kernels: profiles/r06_single_kernel_by_grid.txt "
                         "(the two agree within 4-9 % per level; the trace run is profiled, this one is not)",
        "bytes_per_launch": by, "traffic": traffic,
        "traffic_stale": traffic_stale,
        "traffic_source": "profiles/pmc_summary.json (%s): a committed rocprofv3 --pmc pass of THIS kernel source (SHA-1 of pyramid.hip + "
                          "blur_arith.h checked; stale passes are reported as null), FETCH_SIZE x 2 + WRITE_SIZE per "
                          "MI355X_MICROARCH.md's HBM section; not measured in this run" % traffic_src,
        "measured_copy": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
        "measured_copy_what": "hand-written 16 B/lane copy kernel, 1 GiB read + 1 GiB written (psx_copy_bench)",
        "isolated_replay_avg_ms": round(sum(iso) / len(iso), 5),
        # the other two HBM-bound kernels of octave 0, the same way (stream events around the launch, in the pipeline)
        "level0": {"kernel": "k_level0_x2 (u8 input -> level 0 of octave 0)", "bytes": l0_by, "ms": round(l0_ms, 5),
                   "achieved": round(l0_by / (l0_ms * 1e-3) / 1e9, 1) if l0_ms > 0 else None, "unit": "GB/s",
                   "frac": round(l0_by / (l0_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if l0_ms > 0 else None},
        "extrema": {"kernel": "k_extrema (octave 0: six planes read once, DoG in registers)", "bytes": x0_by, "ms": round(x0_ms, 5),
                    "achieved": round(x0_by / (x0_ms * 1e-3) / 1e9, 1) if x0_ms > 0 else None, "unit": "GB/s",
                    "frac": round(x0_by / (x0_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if x0_ms > 0 else None},
    }

    lap("roofline")
    # ---- one frame at a time on one context (BASELINE config 2, "single frame") ----
    lat = []
    for i in range(25):
        t1 = time.perf_counter()
        c0.extract()
        c0.counts()
        lat.append(time.perf_counter() - t1)
    lat = sorted(lat[5:])
    single_ms = lat[len(lat) // 2] * 1e3
    c0.enable_timers(True)
    c0.extract()
    stages = c0.stage_times()
    c0.enable_timers(False)
    ex["single_frame"] = {"ms": round(single_ms, 4), "value": round(W * H / single_ms / 1e3, 1), "unit": "Mpix/s",
                          "what": "median wall time of one frame at a time on one context (no overlap between frames)"}
    ex["stage_ms_single_frame"] = {"pyramid": round(stages[0], 4), "extrema": round(stages[1], 4),
                                   "orientation": round(stages[2], 4), "descriptors": round(stages[3], 4)}
    # the separable-Gaussian STAGE (every level of every octave, level 0 from the input), not one launch
    stage_b, _ = algorithmic_bytes(W, H, 5)
    stage_gbs = stage_b / (stages[0] * 1e-3) / 1e9
    ex["roofline"]["stage"] = {"bytes": stage_b, "ms": round(stages[0], 4), "achieved": round(stage_gbs, 1), "unit": "GB/s",
                               "frac": round(stage_gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(stage_gbs / copy_gbs, 4),
                               "what": "algorithmic bytes of the whole pyramid build (SURVEY.md 8d: 44 N0 + 48 sum N_o + input) / "
                                       "pyramid stage time of one frame on one context (HIP events around psx_build_pyramid)"}

    # the same stage the way the headline runs it: pipelined.  psx_build_pyramid only (no extrema scan rides along: the
    # interleaving is psx_extract's), all contexts in flight round-robin, device-resident frames
    try:
        nctx = len(ctxs)
        for i, c in enumerate(ctxs):
            c.set_input_tensor(frames[i % len(frames)])
        best = None
        n_pipe = 2400
        for rep in range(2):
            t1 = time.perf_counter()
            for i in range(n_pipe):
                c = ctxs[i % nctx]
                if i >= nctx:
                    c.sync()
                c.build_pyramid()
            for c in ctxs:
                c.sync()
            dtp = time.perf_counter() - t1
            best = dtp if best is None or dtp < best else best
        pipe_gbs = stage_b * n_pipe / best / 1e9
        ex["roofline"]["stage_pipelined"] = {
            "bytes": stage_b, "frames": n_pipe, "contexts": nctx, "ms_per_frame": round(best / n_pipe * 1e3, 4),
            "achieved": round(pipe_gbs, 1), "unit": "GB/s", "frac": round(pipe_gbs / HBM_PEAK_GBS, 4),
            "frac_of_measured_copy": round(pipe_gbs / copy_gbs, 4),
            "what": "the same algorithmic bytes x %d frames / wall time of psx_build_pyramid alone on %d contexts in flight "
                    "(round-robin, each context waits for ITS previous frame only): the regime the headline runs in" % (n_pipe, nctx)}
    except Exception as e:
        ex["roofline"]["stage_pipelined"] = "failed: %s" % e

    lap("single_frame")
    # ---- BASELINE config 5 stand-in and the matcher (always on; a few seconds) ----
    ex["config5"] = config5_leg(capi, np, device)
    lap("config5")
    ex["match"] = match_leg(capi, np, device)
    lap("match")

    # ---- the host side without the kernels: what bends the 1 -> 8 GPU curve (tools/host_ceiling.py) ----
    if world == 1 and not getattr(args, "no_host_ceiling", False):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import host_ceiling
            if args.extras:
                ex["host_ceiling"] = host_ceiling.measure(seconds=1.5, procs=(1, 2, 4, 8), modes=(1, 2))
            else:
                ex["host_ceiling"] = host_ceiling.measure(seconds=1.0, configs=[(2, 8)])      # the PCIe-link rows: --extras, profiles/r05_host_ceiling.json
        except Exception as e:
            ex["host_ceiling"] = "failed: %s" % e

    lap("host_ceiling")
    # ---- every alternative Gauss / scaling / descriptor mode once: single-frame wall time, one context (--extras) ----
    try:
        if not args.extras:
            raise RuntimeError("skipped: informational leg, run with --extras (profiles/ holds the last full line)")
        alt = {}
        for name, kw in (("gauss_relative", dict(gauss_mode=1)), ("gauss_relative_all", dict(gauss_mode=2)),
                         ("gauss_opencv", dict(gauss_mode=3)), ("gauss_fixed9", dict(gauss_mode=4)),
                         ("gauss_fixed15", dict(gauss_mode=5)), ("scale_direct", dict(scaling_mode=0)),
                         ("desc_iloop", dict(desc_mode=1)), ("desc_grid", dict(desc_mode=2)),
                         ("desc_igrid", dict(desc_mode=3)), ("desc_notile", dict(desc_mode=4))):
            ca = capi.Context(capi.default_config(**dict(HEADLINE_KW, **kw)), device=device)
            ca.set_input_tensor(frames[0])
            ts = []
            for i in range(30):                             # 5 warm-up frames (the leg follows the host-ceiling processes), median of 25
                t1 = time.perf_counter()
                ca.extract()
                ca.counts()
                ts.append(time.perf_counter() - t1)
            alt[name] = round(sorted(ts[5:])[len(ts[5:]) // 2] * 1e3, 4)
            ca.close()
        alt["default"] = round(single_ms, 4)
        ex["alt_modes_ms"] = alt
        # GaussMode Fixed9 / Fixed15: the one-kernel octave (pyramid_fixed.hip).  Octave 0 = six planes written from the
        # input image: 24 B per octave-0 pixel + the image, timed like `roofline` (begin / end timestamps of the dispatch
        # inside real psx_extract calls)
        for name, gm in (("fixed9", 4), ("fixed15", 5)):
            cf = capi.Context(capi.default_config(**dict(HEADLINE_KW, gauss_mode=gm)), device=device)
            cf.set_input_tensor(frames[0])
            cf.enable_blur_probe(True)
            acc, nb = 0.0, 0
            for it in range(14):
                cf.extract()
                msf, byf = cf.blur_probe_times()
                if it >= 4 and msf:
                    acc += msf[0]; nb += 1
            cf.enable_timers(True)
            cf.extract()
            stf = cf.stage_times()
            cf.close()
            if nb:
                gbs = byf / (acc / nb * 1e-3) / 1e9
                ex["roofline"][name] = {"kernel": "k_fixed_octave (octave 0, 3840x2160: levels 0..5 from the u8 image in one launch)",
                                        "bytes_per_launch": byf, "avg_launch_ms": round(acc / nb, 5), "achieved": round(gbs, 1), "unit": "GB/s",
                                        "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(gbs / copy_gbs, 4),
                                        "pyramid_stage_ms": round(stf[0], 4)}
    except Exception as e:
        ex["alt_modes_ms"] = str(e) if str(e).startswith("skipped") else "failed: %s" % e

    # ---- BASELINE config 3: 4096x4096, 6 octaves (octave 0 = 8192x8192), one context, device resident (--extras) ----
    try:
        if not args.extras:
            raise RuntimeError("skipped: informational leg, run with --extras (profiles/ holds the last full line)")
        from popsift_amd.synth import synth
        big = torch.from_numpy(np.ascontiguousarray(np.tile(frames_np[0], (4, 3))[:4096, :4096])).to(frames[0].device)
        c3 = capi.Context(capi.default_config(octaves=6, sift_mode=HEADLINE_KW.get("sift_mode", 0)), device=device)
        c3.set_input_tensor(big)
        for _ in range(3):
            c3.extract()
        n3 = c3.counts()
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            c3.extract()
        n3 = c3.counts()
        d3 = (time.perf_counter() - t1) / reps
        # the dominant kernel on THIS configuration's octave 0 (8192 x 8192 planes, 537 MB algorithmic per launch):
        # the same in-pipeline timing as `roofline`, on planes large enough to amortise a launch's ramp
        c3.enable_blur_probe(True)
        acc3, by3 = None, 0.0
        for _ in range(3):
            c3.extract()
            ms3, by3 = c3.blur_probe_times()
            acc3 = ms3 if acc3 is None else [a + b for a, b in zip(acc3, ms3)]
        avg3 = sum(acc3) / 3.0 / len(acc3)
        ex["config3"] = {"value": round(4096 * 4096 / d3 / 1e6, 1), "unit": "Mpix/s", "ms_per_frame": round(d3 * 1e3, 3),
                         "keypoints": n3[0], "descriptors": n3[1],
                         "k_blur_octave0": {"avg_launch_ms": round(avg3, 5), "bytes_per_launch": by3,
                                            "achieved_GBs": round(by3 / (avg3 * 1e-3) / 1e9, 1),
                                            "frac_of_8TBs": round(by3 / (avg3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         "what": "4096x4096 u8 (tiled synthetic frame), octaves=6, upscale x2 (octave 0 = 8192x8192), "
                                 "one context, frames back to back, device resident"}
        c3.close()
        del big
    except Exception as e:                                   # never lose the headline to an extra
        ex["config3"] = str(e) if str(e).startswith("skipped") else "failed: %s" % e

    lap("alt_modes_and_config3")
    # ---- CPU baseline (bounded sample) ----
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        ocfg = po.default_config(**HEADLINE_KW)
        ncores = usable_cores()
        po.run(ocfg, frames_np[0], threads=ncores).close()     # warm
        n_s = 0
        t1 = time.perf_counter()
        while n_s < NBASE and (n_s < 2 or time.perf_counter() - t1 < 6.0):
            po.run(ocfg, frames_np[n_s], threads=ncores).close()
            n_s += 1
        cdt = time.perf_counter() - t1
        strict = round(n_s * W * H / cdt / 1e6, 2)
        # the oracle built -O3 -march=native (contraction allowed) ON THIS HOST is the CPU
        # baseline; the strict IEEE build (-O2 -ffp-contract=off, the parity checker) is stated beside it
        fdt = po.time_fast_build(ocfg, frames_np[:n_s], ncores)
        fast = round(n_s * W * H / fdt / 1e6, 2) if fdt else None
        # value = the faster of the two builds (on the round-4 box the -O2 -mavx2 checker build was the faster one)
        cpu = {"value": max(fast or 0.0, strict), "unit": "Mpix/s", "cores": ncores, "kind": "port",
               "native_build": {"value": fast, "unit": "Mpix/s", "build": "-O3 -march=native -ffp-contract=fast -fopenmp (oracle/_fast, built on this host; None = build failed)"},
               "strict_checker_build": {"value": strict, "unit": "Mpix/s", "build": "-O2 -mavx2 -ffp-contract=off -fopenmp (oracle/liboracle.so)"},
               "sample": "%d frames 1920x1080 (config 1's Config, VLFeat mode), OpenMP over %d threads" % (n_s, ncores)}
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
        ex["cpu_baseline"] = cpu
    else:
        ex["cpu_baseline"] = None
    lap("cpu_baseline")
    ex["extras_wall_s"] = laps
    return ex


if __name__ == "__main__":
    main()
