#!/bin/bash
# dev tool (GPU box): repeated parity runs with different seeds, looking for rare races
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_schedule.py -q -x -p no:cacheprovider 2>&1 | tail -1; done
for seed in 11 22 33 44 55; do timeout 600 python tools/fuzz_sweep.py 200 $seed 1 2>&1 | tail -1; done
for seed in 5 6; do timeout 900 python tools/fuzz_sweep.py 40 $seed 6 2>&1 | tail -1; done
timeout 600 python bench.py --steps 400 --warmup 10 --no-extras --no-cpu-baseline | cut -c1-200
