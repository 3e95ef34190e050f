#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run3; mkdir -p $O
export POPSIFT_HIP_LIB=$GRAFT_REPO_ROOT/popsift_amd/lib_phase/libpopsift_hip.so
for v in "POPSIFT_TILE_NT=1024" "POPSIFT_TILE_NT=512" "POPSIFT_TILE_NT=1024 POPSIFT_TILE_SMALL=0"; do
  echo "== $v"
  env POPSIFT_TILE=1 $v timeout 200 python tools/tile_phase.py 2>&1 | tail -12
done | tee $O/tile_phase.txt
