#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run6; mkdir -p $O
T0=$(date +%s)
python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench default wall: $(( $(date +%s) - T0 )) s"
tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_run6/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "single", d["single_frame"], "stages", d["stage_ms_single_frame"])
print("roofline", {k: v for k, v in d["roofline"].items() if k in ("frac", "avg_launch_ms", "traffic", "traffic_stale", "frac_of_measured_copy", "measured_copy")}, "stage", d["roofline"]["stage"]["frac"], d["roofline"]["stage"]["frac_of_measured_copy"])
print("config5", d["config5"]); print("match", d["match"]); print("host_ceiling", d["host_ceiling"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print("parity", d["parity_checked"]["kp_miss"], d["parity_checked"]["ori_miss"], d["parity_checked"]["desc_miss"], "device_resident", d["device_resident"]["value"])
PY
bash tools/collect_profiles.sh > $O/collect.log 2>&1
tail -5 $O/collect.log
mkdir -p $O/prof; cp -r gpurun_out/prof/* $O/prof/ 2>/dev/null
