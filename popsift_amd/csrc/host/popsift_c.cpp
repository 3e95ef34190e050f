// popsift_c.cpp -- flat C binding of PopSift / SiftJob / FeaturesHost (include/popsift_c.h).
#include "popsift_c.h"
#include "host_pool.h"

#include "popsift/features.h"
#include "popsift/popsift.h"

#include <cstring>
#include <exception>
#include <string>

namespace {
thread_local std::string t_err;

popsift::Config to_config( const psx_config& c )
{
    popsift::Config k;
    k.setOctaves( c.octaves );
    k.setLevels( c.levels );
    k.setSigma( c.sigma );
    k.setEdgeLimit( c.edge_limit );
    k.setThreshold( c.threshold );
    k.setDownsampling( -c.upscale_factor );             // Config::setDownsampling stores -v (sift_conf.cu:235)
    k.setGaussMode( (popsift::Config::GaussMode)c.gauss_mode );
    k.setMode( (popsift::Config::SiftMode)c.sift_mode );
    k.setScalingMode( (popsift::Config::ScalingMode)c.scaling_mode );
    k.setDescMode( (popsift::Config::DescMode)c.desc_mode );
    k.setNormMode( (popsift::Config::NormMode)c.norm_mode );
    k.setNormalizationMultiplier( c.norm_multi );
    if( c.assume_initial_blur ) k.setInitialBlur( c.initial_blur ); else k.setInitialBlur( 0.0f );
    k.setFilterMaxExtrema( c.filter_max_extrema );
    k.setFilterGridSize( c.filter_grid_size );
    k.setFilterSorting( (popsift::Config::GridFilterMode)c.grid_filter_mode );
    return k;
}
} // namespace

extern "C" {

const char* popsift_c_last_error( void ) { return t_err.c_str(); }

popsift_c_handle* popsift_c_create( const psx_config* cfg, int image_mode, int device )
{
    try {
        psx_config d;
        if( cfg == nullptr ) { psx_config_default( &d ); cfg = &d; }
        PopSift* p = new PopSift( to_config( *cfg ), popsift::Config::ExtractingMode,
                                  image_mode ? PopSift::FloatImages : PopSift::ByteImages, device );
        return reinterpret_cast<popsift_c_handle*>( p );
    } catch( const std::exception& e ) { t_err = e.what(); return nullptr; }
}

void popsift_c_destroy( popsift_c_handle* h )
{
    delete reinterpret_cast<PopSift*>( h );              // the destructor calls uninit()
}

popsift_c_job* popsift_c_enqueue_u8( popsift_c_handle* h, int w, int hgt, const unsigned char* img )
{
    try { return reinterpret_cast<popsift_c_job*>( reinterpret_cast<PopSift*>( h )->enqueue( w, hgt, img ) ); }
    catch( const std::exception& e ) { t_err = e.what(); return nullptr; }
}

popsift_c_job* popsift_c_enqueue_f32( popsift_c_handle* h, int w, int hgt, const float* img )
{
    try { return reinterpret_cast<popsift_c_job*>( reinterpret_cast<PopSift*>( h )->enqueue( w, hgt, img ) ); }
    catch( const std::exception& e ) { t_err = e.what(); return nullptr; }
}

popsift_c_features* popsift_c_get( popsift_c_job* job )
{
    SiftJob* j = reinterpret_cast<SiftJob*>( job );
    if( j == nullptr ) return nullptr;
    popsift::FeaturesHost* f = nullptr;
    try { f = j->get(); }
    catch( const std::exception& e ) { t_err = e.what(); f = nullptr; }
    delete j;
    return reinterpret_cast<popsift_c_features*>( f );
}

int popsift_c_feature_count( const popsift_c_features* f )
{
    return f ? reinterpret_cast<const popsift::FeaturesHost*>( f )->getFeatureCount() : -1;
}

int popsift_c_descriptor_count( const popsift_c_features* f )
{
    return f ? reinterpret_cast<const popsift::FeaturesHost*>( f )->getDescriptorCount() : -1;
}

const float* popsift_c_descriptors( const popsift_c_features* f )
{
    if( f == nullptr ) return nullptr;
    popsift::FeaturesHost* fh = const_cast<popsift::FeaturesHost*>( reinterpret_cast<const popsift::FeaturesHost*>( f ) );
    return reinterpret_cast<const float*>( fh->getDescriptors() );
}

int popsift_c_copy( const popsift_c_features* f, psx_feature* features, float* descriptors )
{
    if( f == nullptr ) return PSX_ERR_INVALID;
    popsift::FeaturesHost* fh = const_cast<popsift::FeaturesHost*>( reinterpret_cast<const popsift::FeaturesHost*>( f ) );
    const int ne = fh->getFeatureCount(), no = fh->getDescriptorCount();
    const popsift::Descriptor* base = fh->getDescriptors();
    if( features != nullptr ) {
        const popsift::Feature* src = fh->getFeatures();
        for( int i = 0; i < ne; i++ ) {
            psx_feature& o = features[i];
            o.debug_octave = src[i].debug_octave;
            o.xpos = src[i].xpos; o.ypos = src[i].ypos; o.sigma = src[i].sigma;
            o.num_ori = src[i].num_ori;
            for( int k = 0; k < PSX_ORI_MAX; k++ ) {
                o.orientation[k] = src[i].orientation[k];
                o.desc_idx[k] = src[i].desc[k] ? (int)( src[i].desc[k] - base ) : -1;
            }
        }
    }
    if( descriptors != nullptr && no > 0 ) memcpy( descriptors, base, (size_t)no * sizeof(popsift::Descriptor) );
    return PSX_OK;
}

void popsift_c_free( popsift_c_features* f )
{
    delete reinterpret_cast<popsift::FeaturesHost*>( f );
}

void popsift_c_pool_stats( int device, long long out[6] )
{
    if( out == nullptr ) return;
    const popsift::pool::Stats s = popsift::pool::pinned_stats( device );
    out[0] = s.allocs; out[1] = s.frees; out[2] = s.hits; out[3] = s.free_buffers;
    out[4] = (long long)s.free_bytes; out[5] = (long long)s.in_use;
}

} // extern "C"
