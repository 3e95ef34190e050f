cd $GRAFT_REPO_ROOT
timeout 600 python tools/superframe_ab.py 2>&1 | tail -20
