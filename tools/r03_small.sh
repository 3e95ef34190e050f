#!/bin/bash
# Round 3, experiment 7: one-shot tile kernel (k_blur_small) for the planes that cannot fill the chip.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "^FAILED|^E  " $O/pytest.log | head -8
for v in 0 1 0 1; do
  echo "== POPSIFT_BLUR_SMALL=$v stage ms (pyramid extrema ori desc)"; POPSIFT_BLUR_SMALL=$v timeout 120 python tools/stage_probe.py 2>&1 | tail -1
done | tee $O/stage.log
for v in 0 1; do POPSIFT_BLUR_SMALL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small $v', {k:d[k] for k in ('value','ms_per_step')}, 'dev', d['device_resident']['value'])" | tee -a $O/bench.log; done
cd /tmp && rm -rf /tmp/ps && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python $GRAFT_REPO_ROOT/tools/single_stream.py 20 > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
f = glob.glob("/tmp/ps/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    agg[(n, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k in sorted(agg):
    v = agg[k]; print("%-28s wgs=%6d calls=%4d avg=%8.2f" % (k[0], k[1], len(v), sum(v) / len(v)))
    if k[0].startswith(("k_blur", "k_level0")): tot += sum(v) / 20
print("pyramid kernel time per frame: %.1f us" % tot)
PY
