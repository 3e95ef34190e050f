#!/bin/bash
# usage: tools/pmc_kernel.sh "<counters>" <kernel-substring>   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcx
rocprofv3 --pmc $1 --kernel-trace -d /tmp/pmcx -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/single_stream.py 3 > /tmp/pmcx.log 2>&1
python - "$2" <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open("/tmp/pmcx/r_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[1] not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-30:], int(r["Grid_Size"])//int(r["Workgroup_Size"]))
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg, key=lambda k:-k[1])[:3]:
    print(k, {c: round(sum(v)/len(v)) for c, v in agg[k].items()})
PY
