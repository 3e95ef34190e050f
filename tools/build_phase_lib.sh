#!/bin/bash
# dev tool: build popsift_amd/lib_phase/libpopsift_hip.so with -DPSX_PHASE_TIMING (in-kernel clock64 stamps
# used by tools/blur_phase.py and tools/phase_timing.py); select it with POPSIFT_HIP_LIB=<that .so>
set -e
cd "$(dirname "$0")/.."
mkdir -p popsift_amd/lib_phase
for f in pyramid pyramid_tile pyramid_alt pyramid_fixed pyramid_interp extrema orient_desc gridfilter match util api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPSX_PHASE_TIMING \
      -I include -I popsift_amd/csrc/hip -c popsift_amd/csrc/hip/$f.hip -o popsift_amd/lib_phase/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o popsift_amd/lib_phase/libpopsift_hip.so popsift_amd/lib_phase/*.o
echo built popsift_amd/lib_phase/libpopsift_hip.so
