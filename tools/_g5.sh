R=$GRAFT_REPO_ROOT
cd $R
for mw in 384 520 640 900; do echo "== MINWG=$mw"; POPSIFT_INTERP_MINWG=$mw python tools/fixed_ab.py 1920 1080 20 | grep "relative "; done 2>&1 | tee gpurun_out/interp_minwg.txt
