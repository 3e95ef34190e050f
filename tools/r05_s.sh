cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -k "match" 2>&1 | tail -2
timeout 120 python tools/match_ab.py 2>&1 | tail -2 | cut -c1-200
timeout 120 python tools/match_ab.py 0 2>&1 | tail -2 | cut -c1-200
