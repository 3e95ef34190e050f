#!/bin/bash
# Round 2, GPU call 1: full GPU test suite, bench line, blur A/B, rocprof kernel stats.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02a
rm -rf $OUT; mkdir -p $OUT
cd $R
python -c "import torch" 2>/dev/null
( time timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
for v in "POPSIFT_BLUR_DEFER=1" "POPSIFT_BLUR_DEFER=0" "POPSIFT_BLUR_STEPS=3" "POPSIFT_BLUR_STEPS=4" "POPSIFT_BLUR_STEPS=7" "POPSIFT_BLUR_STEPS=10" "POPSIFT_BLUR_STEPS=17"; do
  env $v timeout 120 python tools/blur_ab.py >> $OUT/blur_ab.jsonl 2>> $OUT/blur_ab.err
done
cat $OUT/blur_ab.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o s -- python $R/tools/single_stream.py 20 > $OUT/single.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $OUT/single_stream_kernel_stats.csv
cp $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) $OUT/single_stream_kernel_trace.csv
head -30 $OUT/single_stream_kernel_stats.csv
