cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "match" 2>&1 | tail -3
python tools/match_ab.py 18432 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
POPSIFT_MATCH_MFMA=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/match_ab.py worker 18432 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $(find /tmp/pm -name "*kernel_trace.csv" | head -1) | grep k_match
