#!/bin/bash
# Round 3, experiment 1: k_blur_dma (LDS-DMA staging, 2 / 3 stage buffers) against the register-staged k_blur.
# Full GPU suite under each variant (planes must stay bit-identical), then in-pipeline per-level timings.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
for v in 3 2 0; do
  POPSIFT_BLUR_DMA=$v timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_dma$v.log 2>&1
  echo "pytest DMA=$v rc=$? $(tail -1 $O/pytest_dma$v.log)"
done
ab() { echo "== $*"; env "$@" timeout 120 python tools/blur_ab.py 2>&1 | tail -1; }
ab POPSIFT_BLUR_DMA=0 | tee -a $O/ab.log
ab POPSIFT_BLUR_DMA=2 | tee -a $O/ab.log
ab POPSIFT_BLUR_DMA=3 | tee -a $O/ab.log
for s in 5 6 7 8; do
  ab POPSIFT_BLUR_DMA=3 POPSIFT_BLUR_DMA_STEPS=$s | tee -a $O/ab.log
  ab POPSIFT_BLUR_DMA=2 POPSIFT_BLUR_DMA_STEPS=$s | tee -a $O/ab.log
done
ab POPSIFT_BLUR_DMA=0 | tee -a $O/ab.log
echo "== 8192^2 planes"
for v in 0 2 3; do echo "DMA=$v"; POPSIFT_BLUR_DMA=$v timeout 200 python tools/blur_ab.py 4096 4096 2>&1 | tail -1 | tee -a $O/ab.log; done
