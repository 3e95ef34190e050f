cd $GRAFT_REPO_ROOT
for hp in 0 1; do echo "== BLUR_HP=$hp"; POPSIFT_BLUR_HP=$hp python tools/stage_times.py 40; POPSIFT_BLUR_HP=$hp python tools/fixed_ab.py 1920 1080 5 | grep default; done
POPSIFT_BLUR_HP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p
POPSIFT_BLUR_HP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o t -- python $GRAFT_REPO_ROOT/tools/single_stream.py 12 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $(find /tmp/p -name "*kernel_trace.csv" | head -1) | grep "k_blur"
