// sift_conf.cpp -- popsift::Config (reference behaviour: sift_conf.cu:18-306)
#include "popsift/sift_conf.h"

#include <sstream>
#include <stdexcept>

namespace popsift {

namespace {
[[noreturn]] void fatal( const char* file, int line, const std::string& msg )
{
    // POP_FATAL convention (common/debug_macros.h:122-127): "file:line\n    message"
    std::ostringstream o;
    o << file << ":" << line << std::endl << "    " << msg;
    throw std::runtime_error( o.str() );
}
} // namespace

#define CONF_FATAL(s) fatal( __FILE__, __LINE__, (s) )

Config::Config( )
    : octaves( -1 )
    , levels( 3 )
    , sigma( 1.6f )
    , _edge_limit( 10.0f )
    , _threshold( 0.04 )
    , _upscale_factor( 1.0f )
    , _log_mode( Config::None )
    , _scaling_mode( Config::ScaleDefault )
    , _desc_mode( Config::Loop )
    , _grid_filter_mode( Config::RandomScale )
    , verbose( false )
    , _max_extrema( 100000 )
    , _filter_max_extrema( -1 )
    , _filter_grid_size( 2 )
    , _gauss_mode( getGaussModeDefault() )
    , _sift_mode( Config::PopSift )
    , _assume_initial_blur( true )
    , _initial_blur( 0.5f )
    , _normalization_mode( getNormModeDefault() )
    , _normalization_multiplier( 0 )
    , _print_gauss_tables( false )
{ }

void Config::setMode( Config::SiftMode m )       { _sift_mode = m; }
void Config::setGaussMode( Config::GaussMode m ) { _gauss_mode = m; }
void Config::setDescMode( Config::DescMode m )   { _desc_mode = m; }

void Config::setDescMode( const std::string& text )
{
    if( text == "loop" )        setDescMode( Config::Loop );
    else if( text == "iloop" )  setDescMode( Config::ILoop );
    else if( text == "grid" )   setDescMode( Config::Grid );
    else if( text == "igrid" )  setDescMode( Config::IGrid );
    else if( text == "notile" ) setDescMode( Config::NoTile );
    else CONF_FATAL( "specified descriptor extraction mode must be one of loop, grid or igrid" );
}

void Config::setGaussMode( const std::string& m )
{
    if( m == "vlfeat" )                       setGaussMode( Config::VLFeat_Compute );
    else if( m == "vlfeat-hw-interpolated" )  setGaussMode( Config::VLFeat_Relative );
    else if( m == "relative" )                setGaussMode( Config::VLFeat_Relative );
    else if( m == "vlfeat-direct" )           setGaussMode( Config::VLFeat_Relative_All );
    else if( m == "opencv" )                  setGaussMode( Config::OpenCV_Compute );
    else if( m == "fixed9" )                  setGaussMode( Config::Fixed9 );
    else if( m == "fixed15" )                 setGaussMode( Config::Fixed15 );
    else CONF_FATAL( std::string("Bad Gauss mode.\n") + getGaussModeUsage() );
}

Config::GaussMode Config::getGaussModeDefault( ) { return Config::VLFeat_Compute; }

const char* Config::getGaussModeUsage( )
{
    return "Choice of Gauss filter method. "
           "Options are: "
           "vlfeat (default), "
           "vlfeat-hw-interpolated, "
           "vlfeat-direct, "
           "opencv, "
           "fixed9, "
           "fixed15, "
           "relative (synonym for vlfeat-hw-interpolated)";
}

bool Config::getCanFilterExtrema() const { return true; }

void Config::setFilterSorting( const std::string& text )
{
    if( text == "up" )          _grid_filter_mode = Config::SmallestScaleFirst;
    else if( text == "down" )   _grid_filter_mode = Config::LargestScaleFirst;
    else if( text == "random" ) _grid_filter_mode = Config::RandomScale;
    else CONF_FATAL( "filter sorting mode must be one of up, down or random" );
}

void Config::setFilterSorting( Config::GridFilterMode m ) { _grid_filter_mode = m; }
void Config::setVerbose( bool on )                        { verbose = on; }
void Config::setLogMode( LogMode mode )                   { _log_mode = mode; }
Config::LogMode Config::getLogMode( ) const               { return _log_mode; }
void Config::setScalingMode( ScalingMode mode )           { _scaling_mode = mode; }

void Config::setUseRootSift( bool on ) { _normalization_mode = on ? RootSift : Classic; }
bool Config::getUseRootSift( ) const   { return ( _normalization_mode == RootSift ); }
Config::NormMode Config::getNormMode( NormMode ) const { return _normalization_mode; }
void Config::setNormMode( Config::NormMode m )         { _normalization_mode = m; }

void Config::setNormMode( const std::string& m )
{
    if( m == "RootSift" )     setNormMode( Config::RootSift );
    else if( m == "classic" ) setNormMode( Config::Classic );
    else CONF_FATAL( std::string("Bad Normalization mode.\n") + getGaussModeUsage() );
}

Config::NormMode Config::getNormModeDefault( ) { return Config::RootSift; }

const char* Config::getNormModeUsage( )
{
    return "Choice of descriptor normalization modes. "
           "Options are: "
           "RootSift (L1-like, default), "
           "Classic (L2-like)";
}

void Config::setNormalizationMultiplier( int mul ) { _normalization_multiplier = mul; }
int  Config::getNormalizationMultiplier( ) const   { return _normalization_multiplier; }

void Config::setDownsampling( float v )   { _upscale_factor = -v; }
void Config::setOctaves( int v )          { octaves = v; }
void Config::setLevels( int v )           { levels = v; }
void Config::setSigma( float v )          { sigma = v; }
void Config::setEdgeLimit( float v )      { _edge_limit = v; }
void Config::setThreshold( float v )      { _threshold = v; }
void Config::setPrintGaussTables()        { _print_gauss_tables = true; }
void Config::setFilterMaxExtrema( int e ) { _filter_max_extrema = e; }
void Config::setFilterGridSize( int sz )  { _filter_grid_size = sz; }

void Config::setInitialBlur( float blur )
{
    _assume_initial_blur = ( blur != 0.0f );
    _initial_blur        = blur;
}

Config::GaussMode Config::getGaussMode( ) const { return _gauss_mode; }
Config::SiftMode  Config::getSiftMode() const   { return _sift_mode; }
bool  Config::hasInitialBlur( ) const           { return _assume_initial_blur; }
float Config::getInitialBlur( ) const           { return _initial_blur; }
float Config::getPeakThreshold() const          { return ( _threshold * 0.5f * 255.0f / levels ); }
bool  Config::ifPrintGaussTables() const        { return _print_gauss_tables; }

bool Config::equal( const Config& o ) const
{
    // the 14 fields the reference compares (sift_conf.cu:286-304)
    return octaves == o.octaves && levels == o.levels && sigma == o.sigma &&
           _edge_limit == o._edge_limit && _threshold == o._threshold &&
           _upscale_factor == o._upscale_factor && _scaling_mode == o._scaling_mode &&
           _max_extrema == o._max_extrema && _gauss_mode == o._gauss_mode &&
           _sift_mode == o._sift_mode && _assume_initial_blur == o._assume_initial_blur &&
           _initial_blur == o._initial_blur && _normalization_mode == o._normalization_mode &&
           _normalization_multiplier == o._normalization_multiplier;
}

} // namespace popsift
