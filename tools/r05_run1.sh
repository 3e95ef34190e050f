#!/bin/bash
# round 5, GPU call 1: the tile kernel -- parity, then schedule A/B, then a kernel trace of the new default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_kernel or pyramid_bit_exact or other_scale" > $O/pytest_tile.log 2>&1
echo "pytest rc=$?" >> $O/pytest_tile.log
tail -5 $O/pytest_tile.log
timeout 900 python tools/sched_ab.py 'POPSIFT_TILE=0' 'POPSIFT_TILE=1' 'POPSIFT_TILE=1 POPSIFT_TILE_TY=32' \
    'POPSIFT_TILE=1 POPSIFT_TILE_NT=1024' 'POPSIFT_TILE=1 POPSIFT_TILE_TY=32 POPSIFT_TILE_NT=1024' \
    'POPSIFT_TILE=1 POPSIFT_TILE_MAXPX=600000' 'POPSIFT_TILE=1 POPSIFT_TILE_MAXPX=9000000' 'POPSIFT_TILE=0' > $O/sched_ab.jsonl 2>&1
cat $O/sched_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    if 'failed' in d: print(d); continue
    print({k: v for k, v in d['env'].items() if 'TILE' in k}, 'single', d['single_ms'], 'stages', d['stage_ms'], 'thr', d['throughput_mpix'], 'kp', d['keypoints'])
"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o s -- python $GRAFT_REPO_ROOT/tools/single_stream.py 20 > $GRAFT_REPO_ROOT/$O/single.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/single_stream_kernel_stats.csv
cp $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) $O/single_kernel_trace.csv
python - <<'PY'
import csv, collections, sys, glob
f = glob.glob('gpurun_out/r05_run1/single_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void (anonymous namespace)::', '')
    wgs = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1) // max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1))
    agg[(name, wgs)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, w), v in sorted(agg.items()):
    print("%-40s wgs=%6d calls=%4d avg=%9.2f min=%9.2f max=%9.2f" % (n[:40], w, len(v), sum(v) / len(v), min(v), max(v)))
PY
