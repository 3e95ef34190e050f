cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -p no:cacheprovider 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do echo BATCH=$v; POPSIFT_BATCH_OCTAVES=$v timeout 120 python tools/latency_probe.py 2>&1 | tail -4; done; done
for v in 0 1; do echo BATCH=$v; POPSIFT_BATCH_OCTAVES=$v timeout 300 python bench.py --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['device_resident']['value'], d['host_export']['value'])"; done
