#!/bin/bash
# usage: tools/pmc_bench.sh "<counters>" [out-file]   (on the GPU box)
# Per-kernel PMC sums over the REAL end-to-end leg of bench.py (C++ API, 8 workers / contexts in flight, --quick: headline leg +
# C-ABI legs), counter pass with --kernel-trace only.  NOTE: rocprofv3's counter collection serialises the dispatches of a
# process (one kernel on the chip at a time): the sums are each kernel's own counters at the headline's workload and launch
# mix, not the counters of kernels overlapping each other.
C=$1; OUT=${2:-/dev/stdout}; case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcb
rocprofv3 --pmc $C --kernel-trace -d /tmp/pmcb -o r --output-format csv -- python $R/bench.py --quick --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-host-ceiling > /tmp/pmcb.log 2>&1
python - "$OUT" <<'PY'
import csv, collections, sys, glob
f = glob.glob("/tmp/pmcb/**/*counter_collection.csv", recursive=True)
if not f: print(open("/tmp/pmcb.log").read()[-3000:]); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
names = set()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names.add(r["Counter_Name"])
    calls[(k, r["Counter_Name"])] += 1
names = sorted(names)
with open(sys.argv[1], "w") as o:
    o.write("# sums over all dispatches of the run: calls " + " ".join(names) + "\n")
    tot = collections.defaultdict(float)
    for k in sorted(agg, key=lambda k: -agg[k][names[0]]):
        o.write("%-42s %7d " % (k, calls[(k, names[0])]) + " ".join("%15.0f" % agg[k][n] for n in names) + "\n")
        for n in names: tot[n] += agg[k][n]
    o.write("%-42s %7s " % ("TOTAL", "") + " ".join("%15.0f" % tot[n] for n in names) + "\n")
PY
tail -2 /tmp/pmcb.log | cut -c1-300
