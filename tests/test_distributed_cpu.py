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

        def e2e_open(self):
            log.append("e2e_open")

        def e2e_enqueue(self, i):
            self.submitted.append(("e2e", i))
            time.sleep(0.0005 * (self.rank + 1))
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


def _worker(rank, world, port, steps, warmup, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import bench
    log = []
    buf = io.StringIO()
    args = bench.parse_args(["--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup)])
    res = bench.run(args, backend_cls=_stub_backend(bench, log), out=buf)
    q.put((rank, res, buf.getvalue(), log))


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
    (_, r0, out0, log0), (_, r1, out1, log1) = res
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
    # whole-job value: all ranks' pixels over the max-over-ranks time (rank 1 is the slower one: 1 ms per frame)
    t = n_frames * bench.W * bench.H / (j["value"] * 1e6)
    assert t >= bench.BATCH * steps * 0.001 * 0.9
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
    assert log0 == log1 == ["e2e_open", "e2e_close", "abi_open", "export False", "export True", "export False", "abi_close"]
