#!/bin/bash
# Round 3: memory-pipe counters of k_blur (TA / TCP / TCC), separate --pmc passes with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "^[[:space:]]*Name:[[:space:]]*[A-Za-z0-9_]*" $O/counters_list.txt | awk '{print $2}' | sort -u | grep -E "^(TA_|TCP_|TCC_|SQ_WAIT|SQ_INST_CYCLES|SQ_ACTIVE_INST|SQ_INSTS_VMEM|SQ_LDS|GRBM_GUI)" | tr '\n' ' ' > $O/counter_names.txt
echo "counters available: $(wc -w < $O/counter_names.txt)"
run() {  # $1 = tag, $2 = counters
  rm -rf /tmp/pmc_$1
  timeout 300 rocprofv3 --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/single_stream.py 3 > /tmp/pmc_$1.log 2>&1
  f=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pass $1 produced nothing: $(tail -2 /tmp/pmc_$1.log)"; return; }
  python - "$f" <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not (name.startswith("k_blur<") or name.startswith("k_level0") or name.startswith("k_extrema") or name.startswith("k_copy")): continue
    wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
    if wgs < 800: continue
    agg[(name, wgs)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("%-34s wgs=%5d " % k + " ".join("%s=%d" % (c, round(sum(v) / len(v))) for c, v in sorted(agg[k].items())))
PY
}
run a "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" | tee $O/pass_a.txt
run b "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" | tee $O/pass_b.txt
run c "TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" | tee $O/pass_c.txt
run d "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" | tee $O/pass_d.txt
run e "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_sum" | tee $O/pass_e.txt
run f "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" | tee $O/pass_f.txt
run g "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" | tee $O/pass_g.txt
