cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run10; mkdir -p $O
T0=$(date +%s)
python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench default wall: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_run10/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "wall", d["wall_s"], "extras", d["extras_wall_s"])
print("match", {k: v for k, v in d["match"].items() if k != "what"})
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_stale"])
PY
