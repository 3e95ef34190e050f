// oracle/ref_shim/sift_config.h -- stands in for the reference's cmake-generated sift_config.h
// (cmake/sift_config.h.in) when its sources are compiled for the CPU.  Test infrastructure.
#pragma once
#define POPSIFT_IS_DEFINED(F) F() == 1
#define POPSIFT_HAVE_SHFL_DOWN_SYNC() 0
#define POPSIFT_HAVE_NORMF()          0
#define POPSIFT_DISABLE_GRID_FILTER() 1   /* s_filtergrid.cu needs Thrust; the filter is off by default */
#define POPSIFT_USE_NVTX()            0
