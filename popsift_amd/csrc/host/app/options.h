// options.h -- the option surface shared by popsift-demo and popsift-match (reference: boost::program_options in
// src/application/main.cpp:49-150 and match.cpp:49-140), re-implemented without Boost: --name value, --name=value,
// boolean switches, and the short forms -h -v -l -i / -l -r.
#pragma once

#include <popsift/sift_conf.h>

#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace app {

struct Option {
    std::string name;         // long name without dashes
    char        shortname;    // 0 = none
    bool        takes_value;
    std::string help;
    std::function<void( const std::string& )> apply;
};

class Options
{
    std::vector<Option> _opts;
public:
    void add( const std::string& name, char s, bool takes_value, const std::string& help, std::function<void( const std::string& )> f )
    {
        _opts.push_back( Option{ name, s, takes_value, help, std::move( f ) } );
    }
    void flag( const std::string& name, char s, const std::string& help, std::function<void()> f )
    {
        add( name, s, false, help, [f]( const std::string& ) { f(); } );
    }
    void usage( std::ostream& o ) const
    {
        o << "Allowed options:" << std::endl;
        for( const Option& op : _opts ) {
            o << "  ";
            if( op.shortname ) o << "-" << op.shortname << " [ --" << op.name << " ]"; else o << "--" << op.name;
            if( op.takes_value ) o << " arg";
            o << std::endl << "        " << op.help << std::endl;
        }
    }
    /// throws std::runtime_error on unknown options / missing values (boost::program_options::error in the reference)
    void parse( int argc, char** argv ) const
    {
        for( int i = 1; i < argc; i++ ) {
            std::string a = argv[i], val;
            const Option* op = nullptr;
            bool have_val = false;
            if( a.size() > 2 && a[0] == '-' && a[1] == '-' ) {
                std::string name = a.substr( 2 );
                const size_t eq = name.find( '=' );
                if( eq != std::string::npos ) { val = name.substr( eq + 1 ); name = name.substr( 0, eq ); have_val = true; }
                for( const Option& o : _opts ) if( o.name == name ) op = &o;
            } else if( a.size() == 2 && a[0] == '-' ) {
                for( const Option& o : _opts ) if( o.shortname == a[1] ) op = &o;
            }
            if( op == nullptr ) throw std::runtime_error( "unrecognised option '" + a + "'" );
            if( op->takes_value && !have_val ) {
                if( i + 1 >= argc ) throw std::runtime_error( "the required argument for option '--" + op->name + "' is missing" );
                val = argv[++i];
            }
            if( !op->takes_value && have_val ) throw std::runtime_error( "option '--" + op->name + "' does not take any arguments" );
            op->apply( val );
        }
    }
};

inline int   to_int( const std::string& s )   { size_t p = 0; int v = std::stoi( s, &p ); if( p != s.size() ) throw std::runtime_error( "bad integer '" + s + "'" ); return v; }
inline float to_float( const std::string& s ) { size_t p = 0; float v = std::stof( s, &p ); if( p != s.size() ) throw std::runtime_error( "bad number '" + s + "'" ); return v; }

/// "Parameters" and "Modes" of the reference's tools (main.cpp:64-123)
inline void add_config_options( Options& o, popsift::Config& config )
{
    o.add( "octaves", 0, true, "Number of octaves", [&]( const std::string& s ) { config.octaves = to_int( s ); } );
    o.add( "levels", 0, true, "Number of levels per octave", [&]( const std::string& s ) { config.levels = to_int( s ); } );
    o.add( "sigma", 0, true, "Initial sigma value", [&]( const std::string& s ) { config.setSigma( to_float( s ) ); } );
    o.add( "threshold", 0, true, "Contrast threshold", [&]( const std::string& s ) { config.setThreshold( to_float( s ) ); } );
    o.add( "edge-threshold", 0, true, "On-edge threshold", [&]( const std::string& s ) { config.setEdgeLimit( to_float( s ) ); } );
    o.add( "edge-limit", 0, true, "On-edge threshold", [&]( const std::string& s ) { config.setEdgeLimit( to_float( s ) ); } );
    o.add( "downsampling", 0, true, "Downscale width and height of input by 2^N", [&]( const std::string& s ) { config.setDownsampling( to_float( s ) ); } );
    o.add( "initial-blur", 0, true, "Assume initial blur, subtract when blurring first time", [&]( const std::string& s ) { config.setInitialBlur( to_float( s ) ); } );
    o.add( "gauss-mode", 0, true, popsift::Config::getGaussModeUsage(), [&]( const std::string& s ) { config.setGaussMode( s ); } );
    o.add( "desc-mode", 0, true, "Choice of descriptor extraction modes: loop, iloop, grid, igrid, notile. Default is loop",
           [&]( const std::string& s ) { config.setDescMode( s ); } );
    o.flag( "popsift-mode", 0, "PopSift extrema refinement and octave-0 shift", [&]() { config.setMode( popsift::Config::PopSift ); } );
    o.flag( "vlfeat-mode", 0, "VLFeat-like extrema refinement (levels remain unchanged)", [&]() { config.setMode( popsift::Config::VLFeat ); } );
    o.flag( "opencv-mode", 0, "OpenCV-like extrema refinement, shift by 0.5, narrower filters", [&]() { config.setMode( popsift::Config::OpenCV ); } );
    o.flag( "direct-scaling", 0, "Direct each octave from upscaled orig instead of blurred level.", [&]() { config.setScalingMode( popsift::Config::ScaleDirect ); } );
    o.add( "norm-multi", 0, true, "Multiply the descriptor by pow(2,<int>).", [&]( const std::string& s ) { config.setNormalizationMultiplier( to_int( s ) ); } );
    o.add( "norm-mode", 0, true, popsift::Config::getNormModeUsage(), [&]( const std::string& s ) { config.setNormMode( s ); } );
    o.flag( "root-sift", 0, popsift::Config::getNormModeUsage(), [&]() { config.setNormMode( popsift::Config::RootSift ); } );
    o.add( "filter-max-extrema", 0, true, "Approximate max number of extrema.", [&]( const std::string& s ) { config.setFilterMaxExtrema( to_int( s ) ); } );
    o.add( "filter-grid", 0, true, "Grid edge length for extrema filtering (ie. value 4 leads to a 4x4 grid)", [&]( const std::string& s ) { config.setFilterGridSize( to_int( s ) ); } );
    o.add( "filter-sort", 0, true, "Sort extrema in each cell by scale, either random (default), up or down", [&]( const std::string& s ) { config.setFilterSorting( s ); } );
    o.flag( "print-gauss-tables", 0, "A debug output printing Gauss filter size and tables", [&]() { config.setPrintGaussTables(); } );
}

} // namespace app
