#!/bin/bash
# Round 3, experiment 4: fused level 0 (k_level0_fused) against k_upscale + k_blur<R,true>; new headline-path tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "passed|failed|Error|error" $O/pytest.log | tail -5
for v in 0 1; do
  echo "== POPSIFT_LEVEL0_FUSED=$v stage ms (pyramid extrema ori desc)"; POPSIFT_LEVEL0_FUSED=$v timeout 120 python tools/stage_probe.py 2>&1 | tail -1
done | tee $O/stage.log
cd /tmp && for v in 0 1; do
  POPSIFT_LEVEL0_FUSED=$v timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_l0_$v -o single -- python $GRAFT_REPO_ROOT/tools/single_stream.py 30 > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$O/prof_l0_$v/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== FUSED=$v $f"; head -14 "$f" | cut -c1-150
done | tee $GRAFT_REPO_ROOT/$O/kernel_stats.txt
