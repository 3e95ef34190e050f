#!/bin/bash
# Round 3: full GPU suite, smoke, default-ish bench run, parity counts.  Outputs under gpurun_out/r03e.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "^FAILED|^ERROR|Error" $O/pytest.log | head -10
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03e/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_frame", "keypoints_per_s", "keypoints_per_1000px", "parity_checked", "sustained", "sparse_frames", "pcie_gbs", "pipe_roofline", "single_frame", "stage_ms_single_frame", "alt_modes_ms"):
    print(k, d.get(k))
print("device_resident", d["device_resident"]["value"], "host_export", d["host_export"]["value"])
print("roofline", {k: v for k, v in d["roofline"].items() if k not in ("traffic_source", "measured_copy_what", "kernel")})
print("config3", d.get("config3")); print("cpu", d.get("cpu_baseline"))
PY
timeout 900 python tools/parity_counts.py 200 $O/parity_counts.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('$O/parity_counts.json')); print({k:(v if k=='per_100k_keypoints' else {q:v[q] for q in ('cases','keypoints','kp_miss','ori_miss','desc_miss','plane_or_extrema_mismatch_cases')}) for k,v in d.items()})"
