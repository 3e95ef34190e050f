#!/bin/bash
# round-5 closing run on the GPU box: full GPU suite, smoke, default bench line (-> profiles/r05_bench_line.json), descriptor SHA
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
T0=$(date +%s)
python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench default wall: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "wall", d["wall_s"])
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_stale"], "stage", d["roofline"]["stage"]["frac"])
print("match", {k: v for k, v in d["match"].items() if k != "what"})
print("config5", {k: v for k, v in d["config5"].items() if k not in ("images", "config")})
print("host_ceiling", d["host_ceiling"]["runs"])
print("parity", d["parity_checked"]["kp_miss"], d["parity_checked"]["ori_miss"], d["parity_checked"]["desc_miss"], "single", d["single_frame"]["ms"], d["stage_ms_single_frame"])
PY
