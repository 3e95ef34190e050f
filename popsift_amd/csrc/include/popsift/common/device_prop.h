// popsift/common/device_prop.h -- device enumeration shim.
// The reference's device_prop_t (common/device_prop.h:23-108) wraps cudaDeviceProp and the CUDA
// texture / surface size limits.  gfx950 has no texture path in this implementation, so the limit
// checks always pass; enumeration, set() and print() are kept because popsift.h exposes the type
// and the demo calls them (main.cpp:298-300).
#pragma once

#include <string>
#include <vector>

namespace popsift {
namespace cuda {

class device_prop_t
{
public:
    enum {
        do_warn   = true,
        dont_warn = false
    };

    device_prop_t();
    ~device_prop_t();

    void print();
    void set(int n, bool print_choice = false);

    bool checkLimit_2DtexLinear(int& width, int& height, bool printWarn) const;
    bool checkLimit_2DtexArray(int& width, int& height, bool printWarn) const;
    bool checkLimit_2DtexLayered(int& width, int& height, int& layers, bool printWarn) const;
    bool checkLimit_2DsurfLayered(int& width, int& height, int& layers, bool printWarn) const;

private:
    struct Info { std::string name; size_t total_mem; int cus; int clock_khz; };
    int               _num_devices;
    std::vector<Info> _properties;
};

} // namespace cuda
} // namespace popsift
