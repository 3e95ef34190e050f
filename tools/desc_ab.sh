#!/bin/bash
# A/B of descriptor kernel variants: rocprofv3 kernel durations on the 1080p bench frame (one context)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pd
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o s -- python $R/tools/single_stream.py 20 > /tmp/pd.log 2>&1
  python - "$v" $(find /tmp/pd -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "k_descriptors" in r["Name"] or "k_orientation" in r["Name"]:
        print(sys.argv[1], r["Name"].split("(anonymous namespace)::")[-1].split("(")[0], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
