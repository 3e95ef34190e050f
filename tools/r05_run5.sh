#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run5; mkdir -p $O
nproc; cat /proc/cpuinfo | grep "model name" | head -1; free -g | head -2
timeout 600 python tools/host_ceiling.py --seconds 2 --procs 1,2,4,8 --modes 1,2 > $O/host_ceiling.json 2> $O/host_ceiling.err
python -c "
import json
d=json.load(open('$O/host_ceiling.json'))
print('cores',d['cores'])
for r in d['runs']: print(r)
"
tail -3 $O/host_ceiling.err
