cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_modes.py -x -q 2>&1 | tail -3
python tools/desc_modes_ms.py 2>&1 | tail -5
