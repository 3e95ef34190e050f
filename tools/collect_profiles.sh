#!/bin/bash
# Run on the GPU box (gpurun -- tools/collect_profiles.sh): regenerates the rocprofv3 evidence that is
# copied into profiles/.  Counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc
# combined with the sys/hip/hsa trace domains).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4
# (the C++ API's worker threads under the profiler crash now and then inside the runtime calls the profiler intercepts -- hipEventQuery, hipStreamSynchronize, a launch; never without the profiler: retry, and do not lose the other passes)
# (round 6: at ~120 k dispatches per second from eight threads the full run died five times out of five on one box; from the third try
# on, the headline leg and the C-ABI legs only -- the same kernels in the same regime, a third of the dispatches)
for try in 1 2 3 4 5 6; do
  QUICK=""; [ $try -ge 3 ] && QUICK="--quick"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python $R/bench.py $QUICK --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-host-ceiling > $OUT/bench.log 2>&1
  if [ -n "$(find /tmp/p1 -name '*kernel_trace.csv' 2>/dev/null | head -1)" ]; then echo "bench trace: try $try, bench.py $QUICK --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-host-ceiling" > $OUT/bench_trace_command.txt; break; fi
  cp $OUT/bench.log $OUT/bench_crash_try$try.log
  rm -rf /tmp/p1
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o s -- python $R/tools/single_stream.py 20 > $OUT/single.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p3 -o f -- python $R/tools/single_stream.py 3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p4 -o w -- python $R/tools/single_stream.py 3 > $OUT/pmc_write.log 2>&1
B1=$(find /tmp/p1 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$B1" ] && cp $B1 $OUT/bench_kernel_stats.csv
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $OUT/single_stream_kernel_stats.csv
BT=$(find /tmp/p1 -name "*kernel_trace.csv" 2>/dev/null | head -1)
python $R/tools/profiles_summarize.py "${BT:-/nonexistent}" $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) \
       $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $(find /tmp/p4 -name "*counter_collection.csv" | head -1) $OUT
bash $R/tools/pmc_desc.sh > /dev/null 2>&1
# which file holds the shader clock on this box (bench.py sustained leg)
{ for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo "== $f"; cat $f; done; rocm-smi --showclocks 2>&1 | head -30; } > $OUT/sclk_probe.txt 2>&1
ls -la $OUT $R/gpurun_out/pmc_desc 2>/dev/null
