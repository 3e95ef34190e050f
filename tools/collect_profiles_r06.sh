#!/bin/bash
# Round 6 additions to tools/collect_profiles.sh (run on the GPU box AFTER it): kernel traces and counters of the rebuilt
# non-default Gauss modes (Fixed9 / Fixed15: pyramid_fixed.hip; VLFeat_Relative: pyramid_interp.hip), the PMC pass over the real
# end-to-end leg.  Output: gpurun_out/prof/r06_*.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{
for m in "gauss_mode=4" "gauss_mode=5" "gauss_mode=1"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python $R/tools/mode_stream.py 12 $m > /dev/null 2>&1
  echo "== rocprofv3 --kernel-trace -- python tools/mode_stream.py 12 $m   (one context, 1080p bench frame; per (kernel, workgroups): calls avg min max us)"
  python $R/tools/trace_by_grid.py $(find /tmp/pm -name "*kernel_trace.csv" | head -1)
done
} > $OUT/r06_modes_kernel_by_grid.txt 2>&1
{
for m in "gauss_mode=4" "gauss_mode=5" "gauss_mode=1"; do
  echo "== $m  (rocprofv3 --pmc, separate passes, python tools/mode_stream.py 4 $m; per (kernel, workgroups) means per dispatch)"
  bash $R/tools/pmc_cmd.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "k_fixed\|k_blur_interp\|k_level0" $R/tools/mode_stream.py 4 $m
  bash $R/tools/pmc_cmd.sh "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "k_fixed\|k_blur_interp\|k_level0" $R/tools/mode_stream.py 4 $m
  bash $R/tools/pmc_cmd.sh "FETCH_SIZE" "k_fixed\|k_blur_interp\|k_level0" $R/tools/mode_stream.py 4 $m
  bash $R/tools/pmc_cmd.sh "WRITE_SIZE" "k_fixed\|k_blur_interp\|k_level0" $R/tools/mode_stream.py 4 $m
done
} > $OUT/r06_modes_pmc.txt 2>&1
bash $R/tools/pmc_bench.sh "SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" $OUT/r06_pmc_bench_sq.txt > /dev/null 2>&1
bash $R/tools/pmc_bench.sh "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_INSTS_SALU SQ_INSTS_LDS" $OUT/r06_pmc_bench_tcc.txt > /dev/null 2>&1
ls -la $OUT | tail -12
