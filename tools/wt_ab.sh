cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 0 1 2; do echo WT=$v; POPSIFT_BLUR_WT=$v timeout 300 python bench.py --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['device_resident']['value'], d['host_export']['value'])"; done; done
