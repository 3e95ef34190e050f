#!/bin/bash
# Round 3, experiment 5: k_blur_hv (H and V pass on different waves, one barrier per step).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
for v in 22 23; do
  POPSIFT_BLUR_DMA=$v timeout 900 python -m pytest tests -m gpu -x -q -k "not headline and not adversarial" > $O/pytest_dma$v.log 2>&1
  echo "pytest DMA=$v rc=$? $(tail -1 $O/pytest_dma$v.log)"
done
ab() { echo "== $*"; env "$@" timeout 120 python tools/blur_ab.py 2>&1 | tail -1; }
for v in 0 22 23; do ab POPSIFT_BLUR_DMA=$v | tee -a $O/ab.log; done
for s in 4 5 7 8; do for v in 22 23; do ab POPSIFT_BLUR_DMA=$v POPSIFT_BLUR_DMA_STEPS=$s | tee -a $O/ab.log; done; done
ab POPSIFT_BLUR_DMA=0 | tee -a $O/ab.log
echo "== 8192^2 planes"
for v in 0 22 23; do echo "DMA=$v"; POPSIFT_BLUR_DMA=$v timeout 200 python tools/blur_ab.py 4096 4096 2>&1 | tail -1 | tee -a $O/ab.log; done
echo "== bench"
for v in 0 22 23; do POPSIFT_BLUR_DMA=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['device_resident']['value'])" | tee -a $O/bench.log; done
