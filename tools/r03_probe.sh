#!/bin/bash
# Round 3, experiment 2: what bounds k_blur?  store width / scope microbenchmark, in-kernel phase stamps and ablations.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 120 tools/ubench/store_width 2>&1 | tee $O/store_width.txt
export POPSIFT_BLUR_DMA=0
export POPSIFT_HIP_LIB=$GRAFT_REPO_ROOT/popsift_amd/lib_phase/libpopsift_hip.so
for dbg in 0 1 2 3 4; do
  echo "== POPSIFT_BLUR_DBG=$dbg (1: loads from 64 cached rows, 2: no stores, 4: stores stay in L2)"
  POPSIFT_BLUR_DBG=$dbg timeout 120 python tools/blur_phase.py 2>&1 | tail -5
done | tee $O/phase.txt
