// popsift/sift_config.h -- build-time feature switches of the MI355X build.
// Counterpart of the reference's generated header (cmake/sift_config.h.in:11-16).
#pragma once

#define POPSIFT_IS_DEFINED(F) F() == 1

#define POPSIFT_HAVE_SHFL_DOWN_SYNC() 0   // CUDA only
#define POPSIFT_HAVE_NORMF()          0   // the non-normf L2 branch is the contract (s_desc_norm_l2.h:86)
#define POPSIFT_DISABLE_GRID_FILTER() 0
#define POPSIFT_USE_NVTX()            0
/* the same host phases are roctx ranges, switched on at run time: POPSIFT_USE_ROCTX=1 (csrc/host/trace.h) */
#define POPSIFT_BACKEND_HIP()         1   // MI355X / gfx950
