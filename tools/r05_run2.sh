#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_kernel or pyramid_bit_exact" > $O/pytest_tile.log 2>&1
echo "pytest rc=$?" >> $O/pytest_tile.log
tail -3 $O/pytest_tile.log
timeout 900 python tools/sched_ab.py 'POPSIFT_TILE=0' 'POPSIFT_TILE=1' 'POPSIFT_TILE=1 POPSIFT_TILE_SMALL=0' 'POPSIFT_TILE=1 POPSIFT_TILE_TY=32' \
    'POPSIFT_TILE=1 POPSIFT_TILE_NT=512' 'POPSIFT_TILE=1 POPSIFT_TILE_TY=32 POPSIFT_TILE_NT=512' \
    'POPSIFT_TILE=1 POPSIFT_TILE_MAXPX=600000' 'POPSIFT_TILE=1 POPSIFT_TILE_MAXPX=200000' 'POPSIFT_TILE=0' > $O/sched_ab.jsonl 2>&1
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
cp $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) $O/single_kernel_trace.csv
python tools/trace_by_grid.py $O/single_kernel_trace.csv | grep -i "tile\|blur"
