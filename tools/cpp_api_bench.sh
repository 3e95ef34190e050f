#!/bin/bash
# dev tool (GPU box): sustained throughput of the C++ API (PopSift::enqueue / SiftJob::get: host images in,
# FeaturesHost out), streaming with at most 16 jobs outstanding; second line: with the grid filter
cd $GRAFT_REPO_ROOT
python - <<'PY'
from popsift_amd.synth import synth
synth(1920,1080,1000).tofile("/tmp/f.raw")
PY
for d in 2 4 8 16; do
  echo "POPSIFT_PIPE_DEPTH=$d"
  POPSIFT_PROFILE=1 POPSIFT_PIPE_DEPTH=$d popsift_amd/lib/popsift-testdriver 1920 1080 /tmp/f.raw /tmp/o.txt --octaves 5 --bench 400 2>&1 | grep -E "bench:|profile"
  POPSIFT_PIPE_DEPTH=$d popsift_amd/lib/popsift-testdriver 1920 1080 /tmp/f.raw /tmp/o.txt --octaves 5 --bench 400 --filter-max 5000 2>&1 | grep -E "bench:"
done
