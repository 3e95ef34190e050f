R=$GRAFT_REPO_ROOT
cd $R
for os in 1 0; do echo "== ONESTEP=$os"; POPSIFT_INTERP_ONESTEP=$os python tools/fixed_ab.py 1920 1080 10 | grep "relative "; done 2>&1 | tee gpurun_out/interp_onestep.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p
rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o t -- python $R/tools/mode_stream.py 8 gauss_mode=1 > /dev/null 2>&1
python $R/tools/trace_by_grid.py $(find /tmp/p -name "*kernel_trace.csv" | head -1) | grep -i "interp\|level0" | tee $R/gpurun_out/interp_trace.txt
