cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -k "match" 2>&1 | tail -3
python tools/match_ab.py 2>&1 | tail -4
python tools/match_ab.py 0 2>&1 | tail -4
