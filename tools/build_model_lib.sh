#!/bin/bash
# dev tool: popsift_amd/lib_model/libpopsift_hip.so = the product objects with orient_desc.hip rebuilt -DPSX_MODEL_NOGRAD
# (tools/polar_patch_model.py); run python -m popsift_amd.build first
set -e
cd "$(dirname "$0")/.."
mkdir -p popsift_amd/lib_model
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPSX_MODEL_NOGRAD -I include -I popsift_amd/csrc/hip \
    -c popsift_amd/csrc/hip/orient_desc.hip -o popsift_amd/lib_model/orient_desc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o popsift_amd/lib_model/libpopsift_hip.so popsift_amd/lib_model/orient_desc.o \
    $(ls popsift_amd/build/*.o | grep -v "host_\|orient_desc.o")
echo built popsift_amd/lib_model/libpopsift_hip.so
