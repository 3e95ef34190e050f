cd $GRAFT_REPO_ROOT
python tools/sched_ab.py 'POPSIFT_REFINE_NT=64' 'POPSIFT_REFINE_NT=128' 'POPSIFT_REFINE_NT=256' 'POPSIFT_REFINE_NT=512' 'POPSIFT_REFINE_NT=1024' 'POPSIFT_REFINE_NT=64' 'POPSIFT_REFINE_NT=512' 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    if 'failed' in d: print(d); continue
    print({k: v for k, v in d['env'].items() if 'REFINE' in k}, 'single', d['single_ms'], 'min', d['single_min_ms'], 'stages', d['stage_ms'], 'thr', d['throughput_mpix'], 'kp', d['keypoints'])
"
cd /tmp && export TMPDIR=/tmp
for nt in 64 512; do
POPSIFT_REFINE_NT=$nt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$nt -- python $GRAFT_REPO_ROOT/tools/single_stream.py 20 > /dev/null 2>&1
echo NT=$nt; grep -h "k_refine\|k_extrema" $(find /tmp/p$nt -name "*kernel_stats.csv") | cut -c1-200
done
