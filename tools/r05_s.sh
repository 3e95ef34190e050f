cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_sweep
timeout 2400 python tools/parity_counts.py 300 gpurun_out/r05_sweep/parity_counts.json 2>&1 | tail -5
