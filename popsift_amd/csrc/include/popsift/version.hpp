// popsift/version.hpp -- version macros (reference: src/popsift/version.hpp)
#pragma once

#define POPSIFT_VERSION_MAJOR 1
#define POPSIFT_VERSION_MINOR 0
#define POPSIFT_VERSION_REVISION 0

#define POPSIFT_TO_STRING_HELPER(x) #x
#define POPSIFT_TO_STRING(x) POPSIFT_TO_STRING_HELPER(x)

#define POPSIFT_VERSION_STRING                                                       \
    POPSIFT_TO_STRING(POPSIFT_VERSION_MAJOR) "." POPSIFT_TO_STRING(POPSIFT_VERSION_MINOR) \
    "." POPSIFT_TO_STRING(POPSIFT_VERSION_REVISION) "-mi355x"
