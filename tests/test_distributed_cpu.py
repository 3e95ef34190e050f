"""world_size-2 gloo test of bench.py's multi-GPU control flow: the real `bench.run()` (rank / world from the
environment, frame i -> GPU i mod N, warm-up, K timed steps between barriers, max-over-ranks time, summed
keypoints, rank 0 prints ONE JSON line) with only the GPU work replaced by a stub backend."""
import io
import json
import os
import socket
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_backend(bench, log):
    class Stub:
        """Same interface as bench.GpuBackend; a frame 'costs' a sleep that differs per rank, and yields a
        keypoint count that is a function of its seed, so the reduced numbers are predictable."""
        dist_backend = "gloo"

        def __init__(self, rank, local_rank, world):
            self.rank, self.world = rank, world
            self.seeds = [bench.frame_seed(j, rank, world) for j in range(bench.BATCH)]
            self.submitted = []
            self.collected = 0
            self.exporting = None

        def sync(self):
            pass

        def reduce_tensor(self, v):
            return torch.tensor([v], dtype=torch.float64)

        def _kp(self, i):
            return self.seeds[i % bench.BATCH] % 7 + 1

        def e2e_open(self, profile="headline"):
            log.append("e2e_open " + profile)

        def e2e_select(self, which):
            log.append("select " + which)

        def e2e_parity(self, indices):
            # rank r "checks" len(indices) frames and reports r + 1 orientation mismatches
            return {"frames": len(indices), "keypoints": 1000 * (self.rank + 1), "descriptors": 1200, "kp_miss": 0,
                    "ori_miss": self.rank + 1, "desc_miss": 0, "max_desc_dist": 1e-4 * (self.rank + 1),
                    "frame_indices": list(indices)}

        def e2e_enqueue(self, i):
            self.submitted.append(("e2e", i))
            time.sleep(0.0001 * (self.rank + 1))
            return i

        def e2e_get(self, job):
            self.collected += 1
            return self._kp(job)

        def e2e_close(self):
            log.append("e2e_close")

        def abi_open(self):
            log.append("abi_open")

        def abi_export(self, on):
            self.exporting = on
            log.append("export %s" % on)

        def abi_submit(self, c, i):
            self.submitted.append(("abi", c, i))

        def abi_collect(self, c):
            self.collected += 1
            return 3

        def abi_close(self):
            log.append("abi_close")

        def extras(self, args, world):
            return {"roofline": None, "cpu_baseline": None}
    return Stub


def _worker(rank, world, port, steps, warmup, q, extra=()):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import bench
    log = []
    buf = io.StringIO()
    args = bench.parse_args(["--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup)] + list(extra))
    stub = _stub_backend(bench, log)
    holder = {}
    orig_init = stub.__init__

    def init(self, *a):
        orig_init(self, *a)
        holder["be"] = self
    stub.__init__ = init
    res = bench.run(args, backend_cls=stub, out=buf)
    q.put((rank, res, buf.getvalue(), log, [t for t in holder["be"].submitted if t[0] == "e2e"][:1] and
           len([t for t in holder["be"].submitted if t[0] == "e2e"])))


@pytest.mark.parametrize("steps,warmup", [(3, 1), (5, 0)])
def test_bench_control_flow_two_ranks(steps, warmup):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, warmup, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, r0, out0, log0, _), (_, r1, out1, log1, _) = res
    # only rank 0 reports, exactly one JSON line
    assert r1 is None and out1 == ""
    lines = [l for l in out0.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j == r0
    import bench
    n_frames = world * bench.BATCH * steps
    assert j["n_gpus"] == world and j["steps"] == steps and j["warmup"] == warmup and j["scaling"] == "weak"
    assert j["config"]["frames_timed"] == n_frames
    # whole-job value: all ranks' pixels over the max-over-ranks time (rank 1 is the slower one: 0.2 ms per frame)
    t = n_frames * bench.W * bench.H / (j["value"] * 1e6)
    assert t >= bench.BATCH * steps * 0.0002 * 0.9
    assert abs(j["ms_per_step"] - t / steps * 1e3) < 0.05 * j["ms_per_step"] + 1e-3
    # keypoints are summed over ranks: every timed frame of both ranks is collected inside the timed region
    seeds = {r: [bench.frame_seed(k, r, world) for k in range(bench.BATCH)] for r in range(world)}
    expect = 0
    for r in range(world):
        for i in range(warmup * bench.BATCH, (warmup + steps) * bench.BATCH):
            expect += seeds[r][i % bench.BATCH] % 7 + 1
    assert abs(j["keypoints_per_frame"] * n_frames - expect) < 0.051 * n_frames
    assert abs(j["device_resident"]["keypoints_per_s"] * (n_frames * bench.W * bench.H / (j["device_resident"]["value"] * 1e6))
               - 3 * n_frames) < 0.02 * 3 * n_frames + 1
    # leg order and the export switch
    assert log0 == log1 == ["e2e_open headline", "select sparse", "select dense", "e2e_close", "e2e_open popsift", "e2e_close",
                            "e2e_open caller", "e2e_close", "abi_open", "export False", "export True", "export False", "abi_close"]
    # parity counts of EVERY rank are summed (rank r reports r + 1 orientation mismatches), the worst distance is the max
    for pc in (j["parity_checked"], j["caller_profile"]["parity_checked"]):
        assert pc["ranks_checked"] == world and pc["frames"] == 4 * world and pc["ori_miss"] == 3
        assert pc["keypoints"] == 3000 and abs(pc["max_desc_dist"] - 2e-4) < 1e-9 and pc["within_budget"] is False
    assert j["popsift_mode"]["value"] > 0 and j["caller_profile"]["value"] > 0 and j["sparse_frames"]["value"] > 0


def test_bench_strong_scaling_shape_two_ranks():
    """--strong = BASELINE config 4's literal shape: BATCH frames per step in TOTAL, BATCH / N per rank; value is still
    all ranks' pixels over the max-over-ranks time."""
    world, steps, warmup = 2, 2, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, warmup, q, ("--strong", "--no-extras"))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import bench
    j = res[0][1]
    assert j["scaling"] == "strong" and j["n_gpus"] == world
    assert j["config"]["frames_per_step_per_gpu"] == bench.BATCH // world
    assert j["config"]["frames_timed"] == bench.BATCH * steps
    # every rank enqueued (warmup + steps) * BATCH / world frames on the end-to-end leg
    for r in res:
        assert r[4] == (warmup + steps) * bench.BATCH // world
