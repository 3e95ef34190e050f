cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['device_resident']['value'], d['host_export']['value'], d['stage_ms_single_frame'], d['single_frame']['ms'])"
