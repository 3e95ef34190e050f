#!/bin/bash
# dev tool (GPU box): repeated parity runs with different seeds, looking for rare races; long throughput run
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak; mkdir -p $O
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_schedule.py tests/test_gpu_headline.py tests/test_gpu_adversarial.py -q -x -p no:cacheprovider 2>&1 | tail -1; done | tee $O/pytest.txt
for seed in 11 22 33 44 55; do timeout 600 python tools/fuzz_sweep.py 200 $seed 1 2>&1 | tail -1; done | tee $O/fuzz.txt
for seed in 5 6; do timeout 900 python tools/fuzz_sweep.py 40 $seed 6 2>&1 | tail -1; done | tee -a $O/fuzz.txt
timeout 600 python bench.py --steps 400 --warmup 10 --no-extras --no-cpu-baseline | cut -c1-400 | tee $O/bench_long.txt
