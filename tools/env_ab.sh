#!/bin/bash
# usage (GPU box): tools/env_ab.sh 'NAME=VALUE ...' 'NAME=VALUE ...'   -- bench.py's timed legs under each runtime environment ('' = default)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  env $v python bench.py --no-extras --steps 10 --warmup 2 2>/dev/null | tail -1 > /tmp/env_ab_line.json
  python - "$v" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/env_ab_line.json"))
    print("%-60s value %8.1f  device_resident %8.1f  host_export %8.1f  pcie %s" % (sys.argv[1] or "(default)", d["value"], d["device_resident"]["value"], d["host_export"]["value"], d.get("pcie_gbs", {}).get("d2h")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
