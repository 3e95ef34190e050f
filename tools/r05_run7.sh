#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
tail -3 gpurun_out/collect.log
ls gpurun_out/prof
