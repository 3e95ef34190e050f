#!/bin/bash
# dev tool: popsift_amd/lib_model*/libpopsift_hip.so = the product objects with orient_desc.hip rebuilt -DPSX_MODEL_NOGRAD / _NOATOMIC / both
# (tools/polar_patch_model.py); run python -m popsift_amd.build first
set -e
cd "$(dirname "$0")/.."
mkdir -p popsift_amd/lib_model
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPSX_MODEL_NOGRAD -I include -I popsift_amd/csrc/hip \
    -c popsift_amd/csrc/hip/orient_desc.hip -o popsift_amd/lib_model/orient_desc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o popsift_amd/lib_model/libpopsift_hip.so popsift_amd/lib_model/orient_desc.o \
    $(ls popsift_amd/build/*.o | grep -v "host_\|orient_desc.o")
for v in NOATOMIC BOTH; do
  mkdir -p popsift_amd/lib_model_$v
  F="-DPSX_MODEL_NOATOMIC"; [ $v = BOTH ] && F="$F -DPSX_MODEL_NOGRAD"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $F -I include -I popsift_amd/csrc/hip \
      -c popsift_amd/csrc/hip/orient_desc.hip -o popsift_amd/lib_model_$v/orient_desc.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o popsift_amd/lib_model_$v/libpopsift_hip.so popsift_amd/lib_model_$v/orient_desc.o \
      $(ls popsift_amd/build/*.o | grep -v "host_\|orient_desc.o")
done
echo built popsift_amd/lib_model/libpopsift_hip.so lib_model_NOATOMIC lib_model_BOTH
