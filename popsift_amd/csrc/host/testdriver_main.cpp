// testdriver_main.cpp -- TEST-ONLY driver over the C++ API on raw frames (no image decoding): used by tests/ and
// tools/cpp_api_bench.sh.  The command line tool with the reference's option surface is app/main.cpp (popsift-demo).
//
// usage: popsift-testdriver <w> <h> <raw-u8-or-f32-file> <out.txt> [--float] [--vlfeat|--opencv]
//                     [--octaves N] [--repeat N] [--norm-multi M] [--classic] [--match <second-raw-file>]
//                     [--filter-max N] grid filter as AliceVision configures it (LargestScaleFirst)
//                     [--bench N]   stream N frames through enqueue/get with at most 16 jobs outstanding per
//                                   device and print the sustained rate (host images in, FeaturesHost out)
//                     [--devices D] with --bench: D PopSift replicas in this process, one per GPU
//                                   (popsift.h:158,166-168); frame i goes to replica i mod D (BASELINE config 4)
// writes: one line per descriptor:  x y sigma orientation d0..d127  (full float precision)
// --match: MatchingMode as in the reference's popsift-match (src/application/match.cpp:257-275): both images
//          are extracted into FeaturesDev objects and lFeatures->match(rFeatures) prints one line per descriptor
#include <popsift/popsift.h>
#include <popsift/features.h>
#include <popsift/sift_conf.h>
#include <popsift/version.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <iostream>
#include <memory>
#include <queue>
#include <vector>

int main( int argc, char** argv )
{
    if( argc < 5 ) { std::cerr << "usage: " << argv[0] << " w h in.raw out.txt [options]" << std::endl; return 2; }
    const int w = atoi( argv[1] ), h = atoi( argv[2] );
    bool is_float = false;
    int repeat = 1;
    const char* match_file = nullptr;
    int bench = 0;
    int devices = 1;
    popsift::Config config;
    for( int i = 5; i < argc; i++ ) {
        if( !strcmp( argv[i], "--float" ) ) is_float = true;
        else if( !strcmp( argv[i], "--vlfeat" ) ) config.setMode( popsift::Config::VLFeat );
        else if( !strcmp( argv[i], "--opencv" ) ) config.setMode( popsift::Config::OpenCV );
        else if( !strcmp( argv[i], "--classic" ) ) config.setNormMode( popsift::Config::Classic );
        else if( !strcmp( argv[i], "--octaves" ) && i + 1 < argc ) config.setOctaves( atoi( argv[++i] ) );
        else if( !strcmp( argv[i], "--norm-multi" ) && i + 1 < argc ) config.setNormalizationMultiplier( atoi( argv[++i] ) );
        else if( !strcmp( argv[i], "--repeat" ) && i + 1 < argc ) repeat = atoi( argv[++i] );
        else if( !strcmp( argv[i], "--match" ) && i + 1 < argc ) match_file = argv[++i];
        else if( !strcmp( argv[i], "--bench" ) && i + 1 < argc ) bench = atoi( argv[++i] );
        else if( !strcmp( argv[i], "--devices" ) && i + 1 < argc ) devices = std::max( 1, atoi( argv[++i] ) );
        else if( !strcmp( argv[i], "--filter-max" ) && i + 1 < argc ) {     // AliceVision: setFilterMaxExtrema + LargestScaleFirst
            config.setFilterMaxExtrema( atoi( argv[++i] ) );
            config.setFilterSorting( popsift::Config::LargestScaleFirst );
        }
    }
    std::vector<unsigned char> raw( (size_t)w * h * ( is_float ? 4 : 1 ) );
    {
        std::ifstream in( argv[3], std::ios::binary );
        if( !in.read( (char*)raw.data(), (std::streamsize)raw.size() ) ) { std::cerr << "short read" << std::endl; return 3; }
    }
    std::cout << "PopSift version: " << POPSIFT_VERSION_STRING << std::endl;

    if( match_file != nullptr ) {
        std::vector<unsigned char> raw2( raw.size() );
        std::ifstream in2( match_file, std::ios::binary );
        if( !in2.read( (char*)raw2.data(), (std::streamsize)raw2.size() ) ) { std::cerr << "short read" << std::endl; return 3; }
        PopSift msift( config, popsift::Config::MatchingMode, is_float ? PopSift::FloatImages : PopSift::ByteImages );
        SiftJob* lJob = is_float ? msift.enqueue( w, h, (const float*)raw.data() ) : msift.enqueue( w, h, raw.data() );
        SiftJob* rJob = is_float ? msift.enqueue( w, h, (const float*)raw2.data() ) : msift.enqueue( w, h, raw2.data() );
        if( !lJob || !rJob ) return 4;
        popsift::FeaturesDev* lFeatures = lJob->getDev();
        popsift::FeaturesDev* rFeatures = rJob->getDev();
        if( !lFeatures || !rFeatures ) return 5;
        std::cout << "Number of features:    " << lFeatures->getFeatureCount() << std::endl;
        std::cout << "Number of descriptors: " << lFeatures->getDescriptorCount() << std::endl;
        std::cout << "Number of features:    " << rFeatures->getFeatureCount() << std::endl;
        std::cout << "Number of descriptors: " << rFeatures->getDescriptorCount() << std::endl;
        std::cout.flush();
        lFeatures->match( rFeatures );
        fflush( stdout );
        delete lFeatures; delete rFeatures; delete lJob; delete rJob;
        msift.uninit();
        return 0;
    }

    if( bench > 0 ) {
        // N replicas, one PopSift per device, results gathered by pointer hand-off only (no device-to-device traffic)
        std::vector<std::unique_ptr<PopSift>> sifts;
        for( int d = 0; d < devices; d++ )
            sifts.emplace_back( new PopSift( config, popsift::Config::ExtractingMode,
                                             is_float ? PopSift::FloatImages : PopSift::ByteImages, d ) );
        std::vector<std::queue<SiftJob*>> q( devices );
        std::vector<size_t> frames_of( devices, 0 );
        size_t kp = 0;
        auto drain_one = [&]( int d ) {
            SiftJob* j = q[d].front(); q[d].pop();
            popsift::Features* fl = j->get();
            if( fl ) { kp += fl->getFeatureCount(); delete fl; }
            delete j;
        };
        auto submit = [&]( int i ) {
            const int d = i % devices;                       // frame i -> GPU i mod N
            SiftJob* j = is_float ? sifts[d]->enqueue( w, h, (const float*)raw.data() ) : sifts[d]->enqueue( w, h, raw.data() );
            if( j ) { q[d].push( j ); frames_of[d]++; }
            if( q[d].size() > 16 ) drain_one( d );
        };
        auto drain_all = [&]() { for( int d = 0; d < devices; d++ ) while( !q[d].empty() ) drain_one( d ); };
        for( int i = 0; i < 32 * devices; i++ ) submit( i );          // warm-up
        drain_all();
        kp = 0;
        std::fill( frames_of.begin(), frames_of.end(), 0 );
        const auto t0 = std::chrono::steady_clock::now();
        for( int i = 0; i < bench; i++ ) submit( i );
        drain_all();
        const double dt = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
        printf( "bench: %d frames %dx%d in %.3f s: %.3f ms/frame, %.0f Mpix/s, %.0f keypoints/frame\n", bench, w, h, dt,
                dt / bench * 1e3, (double)w * h * bench / dt / 1e6, (double)kp / bench );
        if( devices > 1 ) {
            printf( "devices: %d, frames per device:", devices );
            for( int d = 0; d < devices; d++ ) printf( " %zu", frames_of[d] );
            printf( "\n" );
        }
        for( auto& s : sifts ) s->uninit();
        return 0;
    }

    PopSift sift( config, popsift::Config::ExtractingMode, is_float ? PopSift::FloatImages : PopSift::ByteImages );

    std::queue<SiftJob*> jobs;
    for( int r = 0; r < repeat; r++ ) {
        SiftJob* job = is_float ? sift.enqueue( w, h, (const float*)raw.data() ) : sift.enqueue( w, h, raw.data() );
        if( !job ) return 4;
        jobs.push( job );
    }
    int rc = 0;
    bool first = true;
    size_t nf = 0, nd = 0;
    while( !jobs.empty() ) {
        SiftJob* job = jobs.front(); jobs.pop();
        popsift::Features* fl = job->get();
        if( !fl ) { rc = 5; delete job; continue; }
        if( first ) {
            nf = fl->getFeatureCount(); nd = fl->getDescriptorCount();
            std::ofstream of( argv[4] );
            of.precision( 9 );
            for( const popsift::Feature& f : *fl )
                for( int o = 0; o < f.num_ori; o++ ) {
                    of << f.xpos << " " << f.ypos << " " << f.sigma << " " << f.orientation[o];
                    for( int i = 0; i < 128; i++ ) of << " " << f.desc[o]->features[i];
                    of << "\n";
                }
            first = false;
        } else if( (size_t)fl->getFeatureCount() != nf || (size_t)fl->getDescriptorCount() != nd ) {
            std::cerr << "repeat mismatch" << std::endl; rc = 6;
        }
        delete fl;
        delete job;
    }
    std::cerr << "Number of feature points: " << nf << " number of feature descriptors: " << nd << std::endl;
    sift.uninit();
    return rc;
}
