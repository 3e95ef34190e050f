// popsift/sift_constants.h -- public constants (reference: sift_constants.h:35-56).
// The device-side ConstInfo block of the reference lives inside the HIP context here.
#pragma once

#define GAUSS_ALIGN  32
#define GAUSS_LEVELS 12

#define ORI_NBINS     36
#define ORI_WINFACTOR 1.5F

#define DESC_BINS    8
#define DESC_MAGNIFY 3.0f

// VLFeat keeps at most 4 orientations per extremum (Lowe: 3)
#define ORIENTATION_MAX_COUNT 4
