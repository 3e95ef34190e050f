#!/bin/bash
# usage: tools/pmc_frame.sh "<counters>" [out-file]   (on the GPU box)
# Per-kernel PMC sums of ONE 1080p frame (tools/single_stream.py, 3 frames, averaged): where a frame's VALU / SALU /
# LDS instructions go.  Counter passes run with --kernel-trace only (never with sys/hip traces).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcf
rocprofv3 --pmc $1 --kernel-trace -d /tmp/pmcf -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/single_stream.py 3 > /tmp/pmcf.log 2>&1
find /tmp/pmcf -name "*counter_collection.csv" | grep -q . > /dev/null 2>&1 || { tail -20 /tmp/pmcf.log; find /tmp/pmcf | head; exit 1; }
python - "${2:-/dev/stdout}" <<'PY'
import csv, collections, sys
import glob
rows = list(csv.DictReader(open(glob.glob("/tmp/pmcf/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = set()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]) / 3.0
    names.add(r["Counter_Name"])
names = sorted(names)
with open(sys.argv[1], "w") as f:
    f.write("# per frame (3 frames averaged), summed over a kernel's launches: " + " ".join(names) + "\n")
    tot = collections.defaultdict(float)
    for k in sorted(agg, key=lambda k: -agg[k][names[0]]):
        f.write("%-36s " % k + " ".join("%14.0f" % agg[k][n] for n in names) + "\n")
        for n in names: tot[n] += agg[k][n]
    f.write("%-36s " % "TOTAL" + " ".join("%14.0f" % tot[n] for n in names) + "\n")
PY
