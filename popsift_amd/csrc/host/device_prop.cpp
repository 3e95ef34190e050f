// device_prop.cpp -- device enumeration through the C-ABI (no HIP headers on the host side)
#include "popsift/common/device_prop.h"

#include "popsift_hip.h"

#include <iostream>
#include <sstream>
#include <stdexcept>

namespace popsift {
namespace cuda {

device_prop_t::device_prop_t()
    : _num_devices( 0 )
{
    int n = 0;
    if( psx_device_count( &n ) != PSX_OK ) n = 0;
    _num_devices = n;
    for( int i = 0; i < n; i++ ) {
        Info inf;
        char name[256] = {0};
        size_t mem = 0; int cus = 0, clk = 0;
        if( psx_device_info( i, name, sizeof(name), &mem, &cus, &clk ) == PSX_OK ) {
            inf.name = name; inf.total_mem = mem; inf.cus = cus; inf.clock_khz = clk;
        } else {
            inf.name = "?"; inf.total_mem = 0; inf.cus = 0; inf.clock_khz = 0;
        }
        _properties.push_back( inf );
    }
}

device_prop_t::~device_prop_t() = default;

void device_prop_t::print()
{
    for( int i = 0; i < _num_devices; i++ ) {
        const Info& p = _properties[i];
        std::cout << "Device information for device " << i << std::endl
                  << "    Name: " << p.name << std::endl
                  << "    Total global mem: " << p.total_mem << " B" << std::endl
                  << "    Compute units: " << p.cus << std::endl
                  << "    Clock rate: " << p.clock_khz << " kHz" << std::endl;
    }
}

void device_prop_t::set( int n, bool print_choice )
{
    if( n < 0 || n >= _num_devices ) {
        std::ostringstream o;
        o << __FILE__ << ":" << __LINE__ << std::endl
          << "    Runtime error: device " << n << " requested, " << _num_devices << " available";
        throw std::runtime_error( o.str() );
    }
    if( print_choice ) std::cout << "Choosing device " << n << ": " << _properties[n].name << std::endl;
}

// No texture / surface hardware is used on gfx950: the pyramid is plain HBM, so the size limits of
// the reference (device_prop.cu:150-311) do not exist.
bool device_prop_t::checkLimit_2DtexLinear( int&, int&, bool ) const { return true; }
bool device_prop_t::checkLimit_2DtexArray( int&, int&, bool ) const { return true; }
bool device_prop_t::checkLimit_2DtexLayered( int&, int&, int&, bool ) const { return true; }
bool device_prop_t::checkLimit_2DsurfLayered( int&, int&, int&, bool ) const { return true; }

} // namespace cuda
} // namespace popsift
