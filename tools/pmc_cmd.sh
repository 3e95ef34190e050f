#!/bin/bash
# usage: tools/pmc_cmd.sh "<counters>" <kernel-substring> <python script + args ...>   (on the GPU box)
# per (kernel, workgroups): the mean of each counter over the kernel's dispatches.  Counter passes run with --kernel-trace only.
C=$1; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcc
rocprofv3 --pmc $C --kernel-trace -d /tmp/pmcc -o r --output-format csv -- python "$@" > /tmp/pmcc.log 2>&1
python - "$K" <<'PY'
import csv, collections, sys, glob
f = glob.glob("/tmp/pmcc/**/*counter_collection.csv", recursive=True)
if not f: print(open("/tmp/pmcc.log").read()[-2000:]); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    import re
    if not re.search(sys.argv[1].replace("\\|", "|"), r["Kernel_Name"]): continue
    k = (r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ", "").split("(")[0][:40], int(r["Grid_Size"])//int(r["Workgroup_Size"]))
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg, key=lambda k:-k[1]):
    print(k, {c: round(sum(v)/len(v)) for c, v in sorted(agg[k].items())})
PY
