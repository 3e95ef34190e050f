cd /tmp && export TMPDIR=/tmp
for v in V4 V5 V4 V5; do
rm -rf /tmp/pm
POPSIFT_HIP_LIB=$GRAFT_REPO_ROOT/tools/tmp_libs/lib$v.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/match_ab.py worker 18432 > /tmp/pm.log 2>&1
echo "$v rc=$? $(grep seconds /tmp/pm.log | cut -c1-120)"
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/pm/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match_mfma<false>" in r["Name"] or "k_match_exact" in r["Name"]:
            print("    %-22s avg_us %9.2f min %9.2f" % (r["Name"].split("::")[-1][:22], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
