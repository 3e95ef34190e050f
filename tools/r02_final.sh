#!/bin/bash
# Round 2 evidence run: GPU suite, smoke, default bench, rocprofv3 kernel stats and PMC passes (tools/collect_profiles.sh,
# tools/pmc_desc.sh); outputs under gpurun_out/, summaries are copied to profiles/r02_* by hand.
R=$GRAFT_REPO_ROOT
cd $R
bash tools/r02_gpu2.sh
bash tools/collect_profiles.sh > /dev/null 2>&1
bash tools/pmc_desc.sh > /dev/null 2>&1
ls gpurun_out/prof gpurun_out/pmc_desc
