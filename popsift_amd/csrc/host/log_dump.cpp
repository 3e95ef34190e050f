// log_dump.cpp -- Config::LogMode::All debug output (popsift-demo --log).
//
// Restates what the reference writes after every image when logging is on (popsift.cpp:330-338):
//   Pyramid::download_and_save_array -> Octave::download_and_save_array (sift_octave.cu:111-188)
//       dir-octave/pyramid-o-<o>-l-<l>.pgm        Gaussian level, P2, int(v) per pixel            (write_plane2Dunscaled)
//       dir-octave-dump/pyramid-o-<o>-l-<l>.dump  "floats\n<cols> <rows>\n" + raw float32        (dump_plane2Dfloat)
//       dir-dog/d-pyramid-o-<o>-l-<l>.pgm         DoG level, P2, min-max scaled to 0..255          (write_plane2D)
//       dir-dog-txt/d-pyramid-o-<o>-l-<l>.txt     DoG level, P2, int(v) + 127                      (write_plane2Dunscaled)
//       dir-dog-dump/d-pyramid-o-<o>-l-<l>.dump   raw DoG floats
//   Pyramid::save_descriptors (sift_pyramid.cu:88-106) -> writeDescriptor (:401-444)
//       dir-desc/desc-pyramid.txt                 x y sigma orientation[deg] d0..d127
//       dir-fpt/desc-pyramid.txt                  x y sigma orientation[deg]
// These files are what the reference's regression protocol compares byte for byte
// (testScripts/testOxfordDataset.sh.in:65-154).  Formats follow common/write_plane_2d.cu:50-175.
#include "log_dump.h"

#include "popsift/features.h"
#include "popsift/sift_extremum.h"

#include <sys/stat.h>

#include <cmath>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <vector>

namespace popsift {

namespace {

void make_dir( const char* name )
{
    struct stat st;
    if( stat( name, &st ) == -1 ) mkdir( name, 0700 );
}

// write_plane2D (write_plane_2d.cu:50-107): min-max scaled, truncated
void write_plane2D( const std::string& filename, const float* f, int cols, int rows )
{
    float minval = std::numeric_limits<float>::max();
    float maxval = std::numeric_limits<float>::min();
    for( size_t i = 0; i < (size_t)rows * cols; i++ ) { minval = std::min( minval, f[i] ); maxval = std::max( maxval, f[i] ); }
    const float fmaxval = 255.0f / ( maxval - minval );
    std::ofstream of( filename.c_str(), std::ios::binary );
    of << "P2" << std::endl << cols << " " << rows << std::endl << "255" << std::endl;
    for( int row = 0; row < rows; row++ ) {
        for( int col = 0; col < cols; col++ ) {
            const float v = ( f[(size_t)row * cols + col] - minval ) * fmaxval;
            of << (int)(unsigned char)v << " ";
        }
        of << std::endl;
    }
}

// write_plane2Dunscaled (write_plane_2d.cu:110-139): int(v) + offset
void write_plane2Dunscaled( const std::string& filename, const float* f, int cols, int rows, int offset = 0 )
{
    std::ofstream of( filename.c_str(), std::ios::binary );
    of << "P2" << std::endl << cols << " " << rows << std::endl << "255" << std::endl;
    for( int row = 0; row < rows; row++ ) {
        for( int col = 0; col < cols; col++ ) {
            const int val = (int)f[(size_t)row * cols + col];
            of << val + offset << " ";
        }
        of << std::endl;
    }
}

// dump_plane2Dfloat (write_plane_2d.cu:157-175)
void dump_plane2Dfloat( const std::string& filename, const float* f, int cols, int rows )
{
    std::ofstream of( filename.c_str(), std::ios::binary );
    of << "floats" << std::endl << cols << " " << rows << std::endl;
    of.write( (const char*)f, (std::streamsize)( (size_t)rows * cols * sizeof(float) ) );
}

// Pyramid::writeDescriptor (sift_pyramid.cu:401-444), including its second multiplication of the already scaled
// Feature coordinates by 2^(octave - upscale)
void write_descriptor( std::ostream& ostr, const FeaturesHost* features, float up_fac, bool really, bool with_orientation )
{
    if( features->getFeatureCount() == 0 ) return;
    const float M_PI2 = 2.0f * 3.14159265358979323846f;
    FeaturesHost* fh = const_cast<FeaturesHost*>( features );
    for( int ext_idx = 0; ext_idx < features->getFeatureCount(); ext_idx++ ) {
        const Feature& ext = fh->getFeatures()[ext_idx];
        const int   octave = ext.debug_octave;
        const float xpos   = ext.xpos  * pow( 2.0f, octave - up_fac );
        const float ypos   = ext.ypos  * pow( 2.0f, octave - up_fac );
        const float sigma  = ext.sigma * pow( 2.0f, octave - up_fac );
        for( int ori = 0; ori < ext.num_ori; ori++ ) {
            float dom_ori = ext.orientation[ori];
            dom_ori = dom_ori / M_PI2 * 360;
            if( dom_ori < 0 ) dom_ori += 360;
            if( with_orientation )
                ostr << std::setprecision(5) << xpos << " " << ypos << " " << sigma << " " << dom_ori << " ";
            else
                ostr << std::setprecision(5) << xpos << " " << ypos << " " << 1.0f / ( sigma * sigma ) << " 0 "
                     << 1.0f / ( sigma * sigma ) << " ";
            if( really && ext.desc[ori] != nullptr )
                for( float feature : ext.desc[ori]->features ) ostr << feature << " ";
            ostr << std::endl;
        }
    }
}

} // namespace

bool log_dump( psx_ctx* ctx, const FeaturesHost* features, float upscale_factor, const char* basename, std::string* err )
{
    for( const char* d : { "dir-octave", "dir-octave-dump", "dir-dog", "dir-dog-txt", "dir-dog-dump" } ) make_dir( d );
    const int octaves = psx_num_octaves( ctx ), levels = psx_num_levels( ctx );
    std::vector<float> plane;
    for( int o = 0; o < octaves; o++ ) {
        int w = 0, h = 0;
        psx_octave_dims( ctx, o, &w, &h );
        plane.resize( (size_t)w * h );
        for( int l = 0; l < levels; l++ ) {
            if( psx_dump_plane( ctx, PSX_PLANE_GAUSS, o, l, plane.data() ) != PSX_OK ) { if( err ) *err = psx_last_error( ctx ); return false; }
            std::ostringstream a, b;
            a << "dir-octave/" << basename << "-o-" << o << "-l-" << l << ".pgm";
            write_plane2Dunscaled( a.str(), plane.data(), w, h );
            b << "dir-octave-dump/" << basename << "-o-" << o << "-l-" << l << ".dump";
            dump_plane2Dfloat( b.str(), plane.data(), w, h );
        }
        for( int l = 0; l < levels - 1; l++ ) {
            if( psx_dump_plane( ctx, PSX_PLANE_DOG, o, l, plane.data() ) != PSX_OK ) { if( err ) *err = psx_last_error( ctx ); return false; }
            std::ostringstream a, b, c;
            a << "dir-dog/d-" << basename << "-o-" << o << "-l-" << l << ".pgm";
            write_plane2D( a.str(), plane.data(), w, h );
            b << "dir-dog-txt/d-" << basename << "-o-" << o << "-l-" << l << ".txt";
            write_plane2Dunscaled( b.str(), plane.data(), w, h, 127 );
            c << "dir-dog-dump/d-" << basename << "-o-" << o << "-l-" << l << ".dump";
            dump_plane2Dfloat( c.str(), plane.data(), w, h );
        }
    }
    make_dir( "dir-desc" );
    { std::ostringstream n; n << "dir-desc/desc-" << basename << ".txt"; std::ofstream of( n.str().c_str() ); write_descriptor( of, features, upscale_factor, true, true ); }
    make_dir( "dir-fpt" );
    { std::ostringstream n; n << "dir-fpt/desc-" << basename << ".txt"; std::ofstream of( n.str().c_str() ); write_descriptor( of, features, upscale_factor, false, true ); }
    return true;
}

} // namespace popsift
