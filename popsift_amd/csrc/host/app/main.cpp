// popsift-demo -- command line extractor with the reference tool's option surface and outputs
// (src/application/main.cpp:49-329): reads PGM / PPM files (a file or a directory, recursively), runs them through
// PopSift::enqueue / SiftJob::get and writes output-features.txt (x y 1/s^2 0 1/s^2 d0..d127 per descriptor,
// features.cu:310-330); --log adds the dir-octave / dir-dog / dir-desc debug dumps.  No Boost, no DevIL: the
// PGM / PPM reader is the reference's own fallback loader (--pgmread-loading is accepted and is the only loader).
#include "options.h"
#include "pgmread.h"

#include <popsift/common/device_prop.h>
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift/sift_conf.h>
#include <popsift/version.hpp>

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>
#include <list>
#include <memory>
#include <queue>
#include <string>
#include <vector>
#include <cstdlib>

using namespace std;

static bool print_dev_info  = false;
static bool print_time_info = false;
static bool write_as_uchar  = false;
static bool dont_write      = false;
static bool float_mode      = false;
static std::vector<int> device_list;          // one PopSift replica per entry (popsift.h:158,166-168); default: device 0

static void parseargs( int argc, char** argv, popsift::Config& config, string& inputFile )
{
    app::Options all;
    bool help = false;
    all.flag( "help", 'h', "Print usage", [&]() { help = true; } );
    all.flag( "verbose", 'v', "", [&]() { config.setVerbose(); } );
    all.flag( "log", 'l', "Write debugging files", [&]() { config.setLogMode( popsift::Config::All ); } );
    all.add( "input-file", 'i', true, "Input file", [&]( const string& s ) { inputFile = s; } );
    app::add_config_options( all, config );
    all.flag( "print-dev-info", 0, "A debug output printing device information", [&]() { print_dev_info = true; } );
    all.flag( "print-time-info", 0, "A debug output printing image processing time after load()", [&]() { print_time_info = true; } );
    all.flag( "write-as-uchar", 0, "Output descriptors rounded to int. Should be combined with --norm-multi=9 or similar", [&]() { write_as_uchar = true; } );
    all.flag( "dont-write", 0, "Suppress descriptor output", [&]() { dont_write = true; } );
    all.flag( "pgmread-loading", 0, "Use the PGM/PPM loader (the only loader of this build)", []() {} );
    all.flag( "float-mode", 0, "Upload image to GPU as float instead of byte", [&]() { float_mode = true; } );
    // not in the reference tool: several replicas in one process, image i goes to replica i mod N (BASELINE config 4)
    all.add( "devices", 0, true, "Number of GPUs: one PopSift replica on each of devices 0..N-1", [&]( const string& v ) {
        const int n = atoi( v.c_str() );
        if( n < 1 ) throw runtime_error( "--devices needs a positive number" );
        device_list.clear();
        for( int d = 0; d < n; d++ ) device_list.push_back( d );
    } );
    all.add( "device-list", 0, true, "Comma separated device ids, one PopSift replica each (an id may repeat: 0,0)", [&]( const string& v ) {
        device_list.clear();
        size_t pos = 0;
        while( pos <= v.size() ) {
            const size_t c = v.find( ',', pos );
            const string tok = v.substr( pos, c == string::npos ? string::npos : c - pos );
            if( tok.empty() || tok.find_first_not_of( "0123456789" ) != string::npos ) throw runtime_error( "--device-list: bad device id '" + tok + "'" );
            device_list.push_back( atoi( tok.c_str() ) );
            if( c == string::npos ) break;
            pos = c + 1;
        }
    } );
    try {
        all.parse( argc, argv );
        if( help ) { all.usage( cout ); exit( EXIT_SUCCESS ); }
        if( inputFile.empty() ) throw runtime_error( "the option '--input-file' is required but missing" );
    } catch( const std::exception& e ) {
        cerr << "Error: " << e.what() << endl << endl << "Usage:" << endl << endl;
        all.usage( cerr );
        exit( EXIT_FAILURE );
    }
}

static bool is_dir( const string& p )  { struct stat st; return stat( p.c_str(), &st ) == 0 && S_ISDIR( st.st_mode ); }
static bool is_file( const string& p ) { struct stat st; return stat( p.c_str(), &st ) == 0 && S_ISREG( st.st_mode ); }

static void collectFilenames( list<string>& inputFiles, const string& dir )
{
    vector<string> names;
    if( DIR* d = opendir( dir.c_str() ) ) {
        while( dirent* e = readdir( d ) ) {
            const string n = e->d_name;
            if( n != "." && n != ".." ) names.push_back( dir + "/" + n );
        }
        closedir( d );
    }
    sort( names.begin(), names.end() );
    for( const string& p : names ) {
        if( is_file( p ) ) inputFiles.push_back( p );
        else if( is_dir( p ) ) collectFilenames( inputFiles, p );
    }
}

static SiftJob* process_image( const string& inputFile, PopSift& sift )
{
    int w = 0, h = 0;
    unsigned char* image_data = readPGMfile( inputFile, w, h );
    if( image_data == nullptr ) exit( EXIT_FAILURE );
    cout << "Loading " << w << " x " << h << " image " << inputFile << endl;
    SiftJob* job;
    if( !float_mode ) {
        job = sift.enqueue( w, h, image_data );
    } else {
        float* f = new float[(size_t)w * h];
        for( size_t i = 0; i < (size_t)w * h; i++ ) f[i] = float( image_data[i] ) / 256.0f;       // main.cpp:241-245
        job = sift.enqueue( w, h, f );
        delete[] f;
    }
    delete[] image_data;
    return job;
}

static void read_job( SiftJob* job, bool really_write )
{
    popsift::Features* feature_list = job->get();
    cerr << "Number of feature points: " << feature_list->getFeatureCount()
         << " number of feature descriptors: " << feature_list->getDescriptorCount() << endl;
    if( really_write ) {
        ofstream of( "output-features.txt" );
        feature_list->print( of, write_as_uchar );
    }
    delete feature_list;
}

int main( int argc, char** argv )
{
    popsift::Config config;
    list<string> inputFiles;
    string inputFile;

    cout << "PopSift version: " << POPSIFT_VERSION_STRING << endl;
    try {
        parseargs( argc, argv, config, inputFile );
        cout << inputFile << endl;
    } catch( std::exception& e ) {
        cout << e.what() << endl;
        return EXIT_FAILURE;
    }

    if( is_dir( inputFile ) ) {
        cout << inputFile << " is directory" << endl;
        collectFilenames( inputFiles, inputFile );
        if( inputFiles.empty() ) { cerr << "No files in directory, nothing to do" << endl; return EXIT_SUCCESS; }
    } else if( is_file( inputFile ) ) {
        inputFiles.push_back( inputFile );
    } else {
        cout << "Input file is neither regular file nor directory, nothing to do" << endl;
        return EXIT_FAILURE;
    }

    if( device_list.empty() ) device_list.push_back( 0 );
    popsift::cuda::device_prop_t deviceInfo;
    deviceInfo.set( device_list[0], print_dev_info );
    if( print_dev_info ) deviceInfo.print();

    try {
        const auto t0 = chrono::steady_clock::now();
        // One PopSift per listed device (the reference's model for several GPUs: independent objects, popsift.h:158);
        // image i goes to replica i mod N, results are read back in input order.  The replicas share the host: tell
        // them how many they are so that each takes its share of worker threads (POPSIFT_PIPE_DEPTH still overrides).
        const size_t nrep = device_list.size();
        if( nrep > 1 && getenv( "POPSIFT_LOCAL_REPLICAS" ) == nullptr ) setenv( "POPSIFT_LOCAL_REPLICAS", to_string( nrep ).c_str(), 1 );
        vector<unique_ptr<PopSift>> sifts;
        for( int d : device_list )
            sifts.emplace_back( new PopSift( config, popsift::Config::ExtractingMode, float_mode ? PopSift::FloatImages : PopSift::ByteImages, d ) );
        queue<SiftJob*> jobs;
        size_t i = 0;
        for( const string& f : inputFiles ) jobs.push( process_image( f, *sifts[i++ % nrep] ) );
        while( !jobs.empty() ) {
            SiftJob* job = jobs.front(); jobs.pop();
            if( job ) { read_job( job, !dont_write ); delete job; }
        }
        for( auto& sp : sifts ) sp->uninit();
        if( print_time_info )
            cerr << "Processing " << inputFiles.size() << " image(s) took "
                 << chrono::duration<double, milli>( chrono::steady_clock::now() - t0 ).count() << " ms" << endl;
    } catch( const std::exception& e ) {
        cerr << e.what() << endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
