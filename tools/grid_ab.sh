cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_modes.py -q -x -p no:cacheprovider 2>&1 | tail -2
for r in 1 2 3; do timeout 100 python tools/stage_probe.py; timeout 100 python tools/latency_probe.py | tail -1; done
