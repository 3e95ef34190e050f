// popsift/sift_extremum.h -- the SIFT descriptor as returned to callers
// (reference: sift_extremum.h:65-72; InitialExtremum / Extremum are internal to the HIP context).
#pragma once

#include "sift_constants.h"

namespace popsift {

struct Descriptor
{
    float features[128];
};

} // namespace popsift
