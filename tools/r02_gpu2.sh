#!/bin/bash
# Round 2, GPU call: full GPU test suite, smoke, default bench line.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02b
rm -rf $OUT; mkdir -p $OUT
cd $R
python -c "import torch" 2>/dev/null
( time timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; tail -c 4000 $OUT/bench.json; tail -3 $OUT/bench.err
