// popsift-match -- MatchingMode tool with the reference's option surface (src/application/match.cpp:49-300):
// extracts two images into FeaturesDev objects and prints one accept / reject line per left descriptor
// (FeaturesDev::match, features.cu:227-304).
#include "options.h"
#include "pgmread.h"

#include <popsift/common/device_prop.h>
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift/sift_conf.h>
#include <popsift/version.hpp>

#include <sys/stat.h>

#include <iostream>
#include <string>

using namespace std;

static bool print_dev_info = false;

static bool is_file( const string& p ) { struct stat st; return stat( p.c_str(), &st ) == 0 && S_ISREG( st.st_mode ); }

static SiftJob* process_image( const string& inputFile, PopSift& sift )
{
    int w = 0, h = 0;
    unsigned char* image_data = readPGMfile( inputFile, w, h );
    if( image_data == nullptr ) exit( EXIT_FAILURE );
    cout << "Loading " << w << " x " << h << " image " << inputFile << endl;
    SiftJob* job = sift.enqueue( w, h, image_data );
    delete[] image_data;
    return job;
}

int main( int argc, char** argv )
{
    popsift::Config config;
    string lFile, rFile;
    cout << "PopSift version: " << POPSIFT_VERSION_STRING << endl;

    app::Options all;
    bool help = false;
    all.flag( "help", 'h', "Print usage", [&]() { help = true; } );
    all.flag( "verbose", 'v', "", [&]() { config.setVerbose(); } );
    all.flag( "log", 0, "Write debugging files", [&]() { config.setLogMode( popsift::Config::All ); } );
    all.add( "left", 'l', true, "\"Left\"  input file", [&]( const string& s ) { lFile = s; } );
    all.add( "right", 'r', true, "\"Right\" input file", [&]( const string& s ) { rFile = s; } );
    app::add_config_options( all, config );
    all.flag( "print-dev-info", 0, "A debug output printing device information", [&]() { print_dev_info = true; } );
    all.flag( "print-time-info", 0, "accepted for compatibility", []() {} );
    all.flag( "write-as-uchar", 0, "accepted for compatibility", []() {} );
    all.flag( "dont-write", 0, "accepted for compatibility", []() {} );
    all.flag( "pgmread-loading", 0, "Use the PGM/PPM loader (the only loader of this build)", []() {} );
    try {
        all.parse( argc, argv );
        if( help ) { all.usage( cout ); return EXIT_SUCCESS; }
        if( lFile.empty() || rFile.empty() ) throw runtime_error( "the options '--left' and '--right' are required" );
    } catch( const std::exception& e ) {
        cerr << "Error: " << e.what() << endl << endl << "Usage:" << endl << endl;
        all.usage( cerr );
        return EXIT_FAILURE;
    }
    cout << lFile << " <-> " << rFile << endl;
    for( const string& f : { lFile, rFile } )
        if( !is_file( f ) ) { cout << "Input file " << f << " is not a regular file, nothing to do" << endl; return EXIT_FAILURE; }

    popsift::cuda::device_prop_t deviceInfo;
    deviceInfo.set( 0, print_dev_info );
    if( print_dev_info ) deviceInfo.print();

    try {
        PopSift sift( config, popsift::Config::MatchingMode );
        SiftJob* lJob = process_image( lFile, sift );
        SiftJob* rJob = process_image( rFile, sift );
        popsift::FeaturesDev* lFeatures = lJob->getDev();
        cout << "Number of features:    " << lFeatures->getFeatureCount() << endl;
        cout << "Number of descriptors: " << lFeatures->getDescriptorCount() << endl;
        popsift::FeaturesDev* rFeatures = rJob->getDev();
        cout << "Number of features:    " << rFeatures->getFeatureCount() << endl;
        cout << "Number of descriptors: " << rFeatures->getDescriptorCount() << endl;
        cout.flush();
        lFeatures->match( rFeatures );
        fflush( stdout );
        delete lFeatures; delete rFeatures; delete lJob; delete rJob;
        sift.uninit();
    } catch( const std::exception& e ) {
        cerr << e.what() << endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
