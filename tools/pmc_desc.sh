#!/bin/bash
# PMC passes for the gather kernels (k_descriptors / k_orientation): where do the cycles go?
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_desc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm$i -o c -- python $R/tools/single_stream.py 3 > $OUT/run$i.log 2>&1
  f=$(find /tmp/pm$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    wgs=int(r["Grid_Size"])//max(1,int(r["Workgroup_Size"]))
    if n.startswith(("k_descriptors","k_orientation","k_extrema","k_refine")) or (n.startswith("k_blur") and wgs>=900):
        agg[(n,wgs,r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-26s wgs=%5d %-26s %14.0f"%(k[0],k[1],k[2],sum(agg[k])/len(agg[k])))
PY
done
cat $OUT/summary.txt
tail -3 $OUT/run1.log
