// CPU test of the C++ host API (no GPU needed): popsift::Config surface and defaults
// (sift_conf.cu:18-41), string setters and their errors (sift_conf.cu:63-102,132-142,197-203),
// equal() (sift_conf.cu:286-304), PopSift::enqueue mode check (popsift.cpp:247-253) and the
// guarantee that a job is always fulfilled.
#include <popsift/popsift.h>
#include <popsift/features.h>
#include <popsift/sift_conf.h>
#include <popsift/version.hpp>
#include <popsift/sift_config.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <vector>

static int fails = 0;
#define CHECK(c) do { if(!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while(0)

template <class F> static bool throws_runtime_error( F f )
{
    try { f(); } catch( const std::runtime_error& ) { return true; } catch( ... ) { return false; }
    return false;
}

int main()
{
    popsift::Config c;
    CHECK( c.octaves == -1 && c.levels == 3 && c.sigma == 1.6f && c._edge_limit == 10.0f );
    CHECK( c.getUpscaleFactor() == 1.0f && c.getMaxExtrema() == 100000 && c.getFilterMaxExtrema() == -1 );
    CHECK( c.getFilterGridSize() == 2 && c.getGaussMode() == popsift::Config::VLFeat_Compute );
    CHECK( c.getSiftMode() == popsift::Config::PopSift && c.getDescMode() == popsift::Config::Loop );
    CHECK( c.getUseRootSift() && c.getNormalizationMultiplier() == 0 && c.hasInitialBlur() && c.getInitialBlur() == 0.5f );
    CHECK( std::fabs( c.getPeakThreshold() - 0.04f * 0.5f * 255.0f / 3 ) < 1e-6f );
    CHECK( c.getFilterSorting() == popsift::Config::RandomScale && c.getScalingMode() == popsift::Config::ScaleDefault );
    CHECK( c.getLogMode() == popsift::Config::None && !c.verbose );

    c.setDownsampling( -1.0f );  CHECK( c.getUpscaleFactor() == 1.0f );
    c.setDownsampling( 0.0f );   CHECK( c.getUpscaleFactor() == 0.0f );
    c.setGaussMode( "opencv" );  CHECK( c.getGaussMode() == popsift::Config::OpenCV_Compute );
    c.setGaussMode( "relative" );CHECK( c.getGaussMode() == popsift::Config::VLFeat_Relative );
    c.setDescMode( "notile" );   CHECK( c.getDescMode() == popsift::Config::NoTile );
    c.setFilterSorting( "down" );CHECK( c.getFilterSorting() == popsift::Config::LargestScaleFirst );
    c.setNormMode( "classic" );  CHECK( !c.getUseRootSift() );
    c.setNormMode( "RootSift" ); CHECK( c.getUseRootSift() );
    c.setInitialBlur( 0.0f );    CHECK( !c.hasInitialBlur() );
    CHECK( throws_runtime_error( [&]{ c.setGaussMode( "bogus" ); } ) );
    CHECK( throws_runtime_error( [&]{ c.setDescMode( "bogus" ); } ) );
    CHECK( throws_runtime_error( [&]{ c.setFilterSorting( "bogus" ); } ) );
    CHECK( throws_runtime_error( [&]{ c.setNormMode( "bogus" ); } ) );

    popsift::Config a, b;
    CHECK( a == b );
    b.setThreshold( 0.05f ); CHECK( a != b );
    b = a; b.setFilterMaxExtrema( 5 ); CHECK( a == b );     // not among the 14 compared fields
    popart::Config old_spelling; (void)old_spelling;       // README.md:82

    // FeaturesHost ownership and printing
    {
        popsift::FeaturesHost f( 2, 3 );
        CHECK( f.size() == 2 && f.getFeatureCount() == 2 && f.getDescriptorCount() == 3 );
        CHECK( ( (size_t)f.getFeatures() % 4096 ) == 0 && ( (size_t)f.getDescriptors() % 4096 ) == 0 );
        popsift::Feature* e = f.getFeatures();
        for( int i = 0; i < 2; i++ ) {
            e[i].xpos = 1.5f + i; e[i].ypos = 2.5f; e[i].sigma = 2.0f; e[i].num_ori = 1;
            e[i].orientation[0] = 0.f; e[i].desc[0] = f.getDescriptors() + i; e[i].debug_octave = 0;
            for( int k = 0; k < 128; k++ ) e[i].desc[0]->features[k] = 0.25f;
        }
        std::ostringstream o; f.print( o, false );
        CHECK( o.str().find( "1.5 2.5 0.25 0 0.25 " ) == 0 );   // x y 1/s^2 0 1/s^2 ...
        popsift::Features g;  CHECK( g.size() == 0 );
    }

    // PopSift without a usable device: construction works, wrong image mode throws, a job is always
    // fulfilled and get() reports the failure instead of hanging or falling back to a CPU path.
    {
        PopSift ps( popsift::Config(), popsift::Config::ExtractingMode, PopSift::ByteImages );
        std::vector<float> fimg( 64 * 48, 0.5f );
        CHECK( throws_runtime_error( [&]{ ps.enqueue( 64, 48, fimg.data() ); } ) );
        CHECK( ps.testTextureFit( 64, 48 ) == PopSift::Ok );
        CHECK( ps.testTextureFit( 0, 48 ) != PopSift::Ok );
        std::vector<unsigned char> img( 64 * 48, 100 );
        SiftJob* job = ps.enqueue( 64, 48, img.data() );
        CHECK( job != nullptr );
        bool got_error = false, got_features = false;
        try {
            popsift::FeaturesHost* f = job->get();
            got_features = ( f != nullptr );
            delete f;
        } catch( const std::runtime_error& e ) {
            got_error = true;
        }
        // on a GPU box the job succeeds (constant image: zero features); without a GPU it must throw
        CHECK( got_error || got_features );
        const char* has_gpu = std::getenv( "POPSIFT_TEST_EXPECT_GPU" );
        if( has_gpu == nullptr ) { /* either outcome is legal here; see the -m gpu test for the GPU case */ }
        delete job;
        ps.uninit();
    }
    std::printf( "%s version %s\n", fails ? "FAILED" : "ALL OK", POPSIFT_VERSION_STRING );
    return fails ? 1 : 0;
}
