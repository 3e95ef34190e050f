#!/bin/bash
# Round 3, experiment 9: k_blur_dma with ONE stage buffer: the least LDS (27 KB at R <= 8), 5 workgroups per CU.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
POPSIFT_BLUR_DMA=1 timeout 900 python -m pytest tests -m gpu -x -q -k "not headline" > $O/pytest_dma1.log 2>&1; echo "pytest DMA=1 rc=$? $(tail -1 $O/pytest_dma1.log)"
ab() { echo "== $*"; env "$@" timeout 120 python tools/blur_ab.py 2>&1 | tail -1; }
ab POPSIFT_BLUR_DMA=0 | tee -a $O/ab.log
ab POPSIFT_BLUR_DMA=1 | tee -a $O/ab.log
for s in 3 4 5 6; do ab POPSIFT_BLUR_DMA=1 POPSIFT_BLUR_DMA_STEPS=$s | tee -a $O/ab.log; done
ab POPSIFT_BLUR_DMA=0 | tee -a $O/ab.log
for v in 0 1; do POPSIFT_BLUR_DMA=$v POPSIFT_BLUR_DMA_STEPS=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dma $v', {k:d[k] for k in ('value','ms_per_step')}, 'dev', d['device_resident']['value'])" | tee -a $O/bench.log; done
