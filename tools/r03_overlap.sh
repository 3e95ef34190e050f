#!/bin/bash
# Round 3, experiment 6: does leaving LDS room on the CUs (fewer resident k_blur workgroups) raise the throughput
# of 16 contexts (kernels of different frames sharing CUs)?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
for pad in 0 12288 24576 0; do
  POPSIFT_BLUR_LDS_PAD=$pad timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', {k:d[k] for k in ('value','ms_per_step')}, 'dev', d['device_resident']['value'], 'export', d['host_export']['value'])" | tee -a $O/bench.log
done
