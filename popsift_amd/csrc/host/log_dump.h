// log_dump.h -- Config::LogMode::All debug output (internal to libpopsift)
#pragma once

#include "popsift_hip.h"

#include <string>

namespace popsift {
class FeaturesHost;
/// Writes the reference's dir-octave / dir-dog / dir-desc / dir-fpt debug files for the frame the context has just
/// extracted (popsift.cpp:330-338).  Returns false and fills *err on failure.
bool log_dump( psx_ctx* ctx, const FeaturesHost* features, float upscale_factor, const char* basename, std::string* err );
} // namespace popsift
