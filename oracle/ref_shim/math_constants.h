// oracle/ref_shim/math_constants.h -- CUDA's math_constants.h for the CPU emulation (test infrastructure)
#pragma once
#include <cmath>
#include <limits>
#define CUDART_INF_F      (std::numeric_limits<float>::infinity())
#define CUDART_NAN_F      (std::numeric_limits<float>::quiet_NaN())
#define CUDART_MAX_NORMAL_F (std::numeric_limits<float>::max())
#define CUDART_PI_F       3.141592654f
