// popsift/sift_conf.h -- popsift::Config, the tunables of the extraction algorithm.
//
// Same public surface as the reference (sift_conf.h:29-421): enums, setters (typed and by
// string), getters, public data members octaves / levels / sigma / _edge_limit / verbose,
// equal() and the comparison operators, so callers such as AliceVision compile unchanged.
// Unlike the reference's constructor (sift_conf.cu:42-50) this one never touches a device.
#pragma once

#include <string>

#define MAX_OCTAVES   20
#define MAX_LEVELS    10

#ifdef _MSC_VER
#define DEPRECATED(func) __declspec(deprecated) func
#elif defined(__GNUC__) || defined(__clang__)
#define DEPRECATED(func) func __attribute__ ((deprecated))
#else
#define DEPRECATED(func) func
#endif

namespace popsift {

struct Config
{
    Config();

    /// how the 1-D Gaussian tables are built
    enum GaussMode { VLFeat_Compute, VLFeat_Relative, VLFeat_Relative_All, OpenCV_Compute, Fixed9, Fixed15 };
    /// which other SIFT implementation the details mimic
    enum SiftMode { PopSift, OpenCV, VLFeat, Default = PopSift };
    enum LogMode { None, All };
    enum ScalingMode { ScaleDirect, ScaleDefault };
    /// descriptor extraction variants
    enum DescMode { Loop, ILoop, Grid, IGrid, NoTile };
    /// descriptor normalisation
    enum NormMode { RootSift, Classic };
    /// which extrema the grid filter keeps
    enum GridFilterMode { RandomScale, LargestScaleFirst, SmallestScaleFirst };
    /// what is kept after processing
    enum ProcessingMode { ExtractingMode, MatchingMode };

    void setGaussMode( const std::string& m );
    void setGaussMode( GaussMode m );
    void setMode( SiftMode m );
    void setLogMode( LogMode mode = All );
    void setScalingMode( ScalingMode mode = ScaleDefault );
    void setVerbose( bool on = true );
    void setDescMode( const std::string& byname );
    void setDescMode( DescMode mode = Loop );

    void setDownsampling( float v );
    void setOctaves( int v );
    void setLevels( int v );
    void setSigma( float v );
    void setEdgeLimit( float v );
    void setThreshold( float v );
    void setInitialBlur( float blur );
    void setPrintGaussTables( );
    void setFilterMaxExtrema( int extrema );
    void setFilterGridSize( int sz );
    void setFilterSorting( const std::string& direction );
    void setFilterSorting( GridFilterMode m );

    bool  hasInitialBlur( ) const;
    float getInitialBlur( ) const;

    /// threshold * 0.5 * 255 / levels
    float getPeakThreshold() const;

    bool ifPrintGaussTables() const;
    GaussMode getGaussMode( ) const;
    static GaussMode getGaussModeDefault( );
    static const char* getGaussModeUsage( );
    SiftMode getSiftMode() const;
    LogMode getLogMode() const;

    /// number of octaves; -1 = derive from the image size
    int      octaves;
    /// searchable DoG levels per octave (Gaussian levels = levels + 3)
    int      levels;
    float    sigma;
    float    _edge_limit;

    void               setNormMode( NormMode m );
    void               setNormMode( const std::string& m );
    DEPRECATED(void    setUseRootSift( bool on ));
    bool               getUseRootSift( ) const;
    NormMode           getNormMode( NormMode m ) const;
    NormMode           getNormMode( ) const { return _normalization_mode; }
    static NormMode    getNormModeDefault( );
    static const char* getNormModeUsage( );

    int  getNormalizationMultiplier( ) const;
    void setNormalizationMultiplier( int mul );

    /// the input image is stretched by 2^upscale_factor before processing
    inline float getUpscaleFactor( ) const { return _upscale_factor; }
    int getMaxExtrema( ) const { return _max_extrema; }
    bool getCanFilterExtrema() const;
    int getFilterMaxExtrema() const { return _filter_max_extrema; }
    int getFilterGridSize() const { return _filter_grid_size; }
    GridFilterMode getFilterSorting() const { return _grid_filter_mode; }
    inline ScalingMode getScalingMode() const { return _scaling_mode; }
    inline DescMode getDescMode() const { return _desc_mode; }
    float getThreshold() const { return _threshold; }

    bool equal( const Config& other ) const;

private:
    float          _threshold;
    float          _upscale_factor;
    LogMode        _log_mode;
    ScalingMode    _scaling_mode;
    DescMode       _desc_mode;
    GridFilterMode _grid_filter_mode;

public:
    bool     verbose;

private:
    int       _max_extrema;
    int       _filter_max_extrema;
    int       _filter_grid_size;
    GaussMode _gauss_mode;
    SiftMode  _sift_mode;
    bool      _assume_initial_blur;
    float     _initial_blur;
    NormMode  _normalization_mode;
    int       _normalization_multiplier;
    bool      _print_gauss_tables;
};

inline bool operator==( const Config& l, const Config& r ) { return l.equal( r ); }
inline bool operator!=( const Config& l, const Config& r ) { return ! l.equal( r ); }

} // namespace popsift

// old spelling used in the reference's README (README.md:82)
namespace popart = popsift;
