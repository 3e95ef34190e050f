cd $GRAFT_REPO_ROOT
export POPSIFT_HIP_LIB=$GRAFT_REPO_ROOT/popsift_amd/lib_phase/libpopsift_hip.so
for d in 0 3; do echo "DBG=$d"; POPSIFT_BLUR_DBG=$d timeout 120 python tools/blur_phase.py; done
