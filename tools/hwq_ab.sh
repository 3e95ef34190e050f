cd $GRAFT_REPO_ROOT
for q in 0 2 4 8 16 24; do echo HWQ=$q; if [ $q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi; timeout 300 python bench.py --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['device_resident']['value'], d['host_export']['value'])"; done
