#!/bin/bash
# usage: tools/pmc_one.sh <kernel-substring> [ENV=VAL ...]   PMC counters of one kernel on the 1080p frame
R=$GRAFT_REPO_ROOT
K=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  rm -rf /tmp/pq
  env "$@" timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pq -o c -- python $R/tools/single_stream.py 3 > /tmp/pq.log 2>&1
  python - "$(find /tmp/pq -name '*counter_collection.csv' | head -1)" "$K" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    if sys.argv[2] in n: agg[(n,r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-18s %-24s %14.0f"%(k[0],k[1],sum(agg[k])/len(agg[k])))
PY
done
