// trace.h -- optional profiler ranges.  The reference brackets its host phases with NVTX ranges behind the build
// flag POPSIFT_USE_NVTX (popsift.cpp:441-452 "inserting image", sift_pyramid.cu:288-319 "download descriptors");
// here the same phases are roctx ranges (visible in rocprofv3 --marker-trace), switched on at RUN time with
// POPSIFT_USE_ROCTX=1: libroctx64.so is dlopen'ed on first use, so the library has no link-time dependency on it
// and costs one predictable branch per range when the switch is off.
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace popsift {
namespace trace {

struct Api {
    int  (*push)(const char*) = nullptr;
    int  (*pop)() = nullptr;
    bool on = false;
};

inline const Api& api()
{
    static const Api a = [] {
        Api v;
        const char* e = getenv("POPSIFT_USE_ROCTX");
        if (e == nullptr || e[0] != '1') return v;
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) return v;
        v.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        v.pop  = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        v.on = v.push != nullptr && v.pop != nullptr;
        return v;
    }();
    return a;
}

// scoped range on the calling thread
struct Range {
    bool active;
    explicit Range(const char* name) : active(api().on) { if (active) api().push(name); }
    ~Range() { if (active) api().pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

inline bool enabled() { return api().on; }

} // namespace trace
} // namespace popsift
