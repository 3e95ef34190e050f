// ten-line consumer of the drop-in: the includes and calls an AliceVision-style caller makes
#include <popsift/popsift.h>
#include <popsift/features.h>
#include <popsift/sift_conf.h>
#include <popsift/version.hpp>
#include <iostream>
int main()
{
    popsift::Config config;
    config.setOctaves( 4 ); config.setLevels( 3 ); config.setDownsampling( -1 ); config.setThreshold( 0.04f );
    config.setEdgeLimit( 10.0f ); config.setNormalizationMultiplier( 9 ); config.setNormMode( popsift::Config::RootSift );
    config.setFilterMaxExtrema( 5000 ); config.setFilterSorting( popsift::Config::LargestScaleFirst );
    std::cout << "consumer linked against PopSift " << POPSIFT_VERSION_STRING << ", peak threshold " << config.getPeakThreshold() << std::endl;
    popsift::FeaturesHost f( 3, 4 );                     // host-only part of the API: works without a GPU
    return ( f.getFeatureCount() == 3 && f.getDescriptorCount() == 4 ) ? 0 : 1;
}
