// oracle/ref_shim/sift_config.h -- stands in for the reference's cmake-generated sift_config.h
// (cmake/sift_config.h.in) when its sources are compiled for the CPU.  Test infrastructure.
#pragma once
#define POPSIFT_IS_DEFINED(F) F() == 1
#define POPSIFT_HAVE_SHFL_DOWN_SYNC() 0
#define POPSIFT_HAVE_NORMF()          0
/* s_filtergrid.cu needs Thrust: the Makefile compiles it against rocThrust's headers with the serial
 * CPP device system (SHIM_HAVE_THRUST); without them the reference's own disabled branch is built. */
#ifdef SHIM_HAVE_THRUST
#define POPSIFT_DISABLE_GRID_FILTER() 0
#else
#define POPSIFT_DISABLE_GRID_FILTER() 1
#endif
#define POPSIFT_USE_NVTX()            0
