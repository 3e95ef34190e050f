cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -q -x -p no:cacheprovider 2>&1 | tail -2
for r in 1 2 3; do timeout 120 python tools/latency_probe.py 2>&1 | tail -1; timeout 120 python tools/blur_ab.py | cut -c1-200; done
