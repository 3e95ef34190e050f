#!/bin/bash
# round-6 closing run on the GPU box: profiles (tools/collect_profiles.sh + tools/collect_profiles_r06.sh), the default bench line,
# the --extras line, smoke
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
bash tools/collect_profiles.sh > $O/collect.log 2>&1
bash tools/collect_profiles_r06.sh > $O/collect_r06.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
T0=$(date +%s)
python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench default wall: $(( $(date +%s) - T0 )) s"
python bench.py --extras > $O/bench_line_extras.json 2> $O/bench_extras.err
python - <<'PY'
import json
for f in ("bench_line.json", "bench_line_extras.json"):
    d = json.loads(open("gpurun_out/r06_final/" + f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value", d["value"], "wall", d["wall_s"])
    print("  roofline", r["frac"], r["avg_launch_ms"], "traffic", r["traffic"], r["traffic_stale"], "stage", r["stage"]["frac"], "pipelined", r["stage_pipelined"]["frac"], r["stage_pipelined"]["frac_of_measured_copy"])
    print("  fixed9", r.get("fixed9"), "\n  fixed15", r.get("fixed15"))
    print("  alt", d.get("alt_modes_ms"))
    print("  match", {k: v for k, v in d["match"].items() if k != "what"})
    print("  parity", d["parity_checked"]["kp_miss"], d["parity_checked"]["ori_miss"], d["parity_checked"]["desc_miss"], "single", d["single_frame"]["ms"], d["stage_ms_single_frame"])
PY
