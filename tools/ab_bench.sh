#!/bin/bash
# usage (GPU box): tools/ab_bench.sh ab/<variant> [ab/<variant> ...]
# A/B of kernel variants ON ONE BOX: boxes differ by +-3 % in bench.py's value, more than most single changes are worth.
# Build each variant in the container (python -m popsift_amd.build), copy popsift_amd/lib/libpopsift_hip.so to
# ab/<variant>/ (ab/ travels with the snapshot; do not commit it), then run this with the variants listed twice in
# alternating order.  Prints value / sparse / caller / device-resident / single-frame per run.
cd $GRAFT_REPO_ROOT
cp popsift_amd/lib/libpopsift_hip.so /tmp/libpopsift_hip.keep
for v in "$@"; do
    cp $v/libpopsift_hip.so popsift_amd/lib/libpopsift_hip.so
    python bench.py --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print(sys.argv[1], d["value"], {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.items()
      if k in ("sparse_frames", "caller_profile", "device_resident", "single_frame")}, "k_blur frac", d["roofline"]["frac"],
      "stages", d["stage_ms_single_frame"])
PY
done
cp /tmp/libpopsift_hip.keep popsift_amd/lib/libpopsift_hip.so
