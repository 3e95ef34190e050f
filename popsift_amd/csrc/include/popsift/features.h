// popsift/features.h -- result types handed to the caller.
//
// Same public surface as the reference (features.h:23-122): Feature (coordinates in input-image
// pixels, up to 4 orientations, pointers into the owning object's descriptor array),
// FeaturesBase, FeaturesHost (alias Features) and FeaturesDev.
#pragma once

#include "sift_constants.h"

#include <iostream>
#include <vector>

namespace popsift {

struct Descriptor; // float features[128]

struct Feature
{
    int         debug_octave;
    float       xpos;
    float       ypos;
    /// scale
    float       sigma;
    /// number of valid entries in orientation[] / desc[]
    int         num_ori;
    float       orientation[ORIENTATION_MAX_COUNT];
    Descriptor* desc[ORIENTATION_MAX_COUNT];

    void print( std::ostream& ostr, bool write_as_uchar ) const;
};

std::ostream& operator<<( std::ostream& ostr, const Feature& feature );

class FeaturesBase
{
    int _num_ext;
    int _num_ori;

public:
    FeaturesBase( );
    virtual ~FeaturesBase( );

    inline int  size() const                { return _num_ext; }
    inline int  getFeatureCount() const     { return _num_ext; }
    inline int  getDescriptorCount() const  { return _num_ori; }

    inline void setFeatureCount( int num_ext )    { _num_ext = num_ext; }
    inline void setDescriptorCount( int num_ori ) { _num_ori = num_ori; }
};

/// Host-resident result: arrays owned by the object.  Objects produced by PopSift hold pooled buffers
/// (the descriptor array is the pinned buffer the GPU wrote the descriptors into: no copy); they go back
/// to a process-wide pool when the object is deleted.
class FeaturesHost : public FeaturesBase
{
    Feature*     _ext;
    Descriptor*  _ori;
    size_t       _ext_cap;   // bytes; 0: _ext came from posix_memalign
    size_t       _ori_cap;   // bytes; 0: _ori came from posix_memalign

public:
    FeaturesHost( );
    FeaturesHost( int num_ext, int num_ori );
    ~FeaturesHost( ) override;

    typedef Feature*       F_iterator;
    typedef const Feature* F_const_iterator;

    inline F_iterator       begin()       { return _ext; }
    inline F_const_iterator begin() const { return _ext; }
    inline F_iterator       end()         { return &_ext[size()]; }
    inline F_const_iterator end() const   { return &_ext[size()]; }

    void reset( int num_ext, int num_ori );
    /// kept for source compatibility: results arrive in pooled pinned buffers (DMA download or zero-copy export), nothing to pin per image
    void pin( );
    void unpin( );

    inline Feature*    getFeatures()    { return _ext; }
    inline Descriptor* getDescriptors() { return _ori; }

    void print( std::ostream& ostr, bool write_as_uchar ) const;

    /// internal (PopSift): take ownership of pooled buffers (see host_pool.h); caps in bytes
    void adopt( int num_ext, int num_ori, Feature* ext, size_t ext_cap, Descriptor* ori, size_t ori_cap );

protected:
    friend class Pyramid;
    void release( );
};

using Features = FeaturesHost;

std::ostream& operator<<( std::ostream& ostr, const FeaturesHost& feature );

/// Device-resident result (MatchingMode): arrays live in HBM of the extracting device.
class FeaturesDev : public FeaturesBase
{
    Feature*     _ext;   // device: psx_feature records (indices instead of pointers)
    Descriptor*  _ori;   // device
    int*         _rev;   // device: descriptor -> extremum
    int          _device;

public:
    FeaturesDev( );
    FeaturesDev( int num_ext, int num_ori );
    ~FeaturesDev( ) override;

    void reset( int num_ext, int num_ori );

    /// brute-force 2-NN matcher of the reference (features.cu:160-304): prints one accept/reject line
    /// per descriptor of *this, as the reference's show_distance does
    void match( FeaturesDev* other );

    inline Feature*    getFeatures()    { return _ext; }
    inline Descriptor* getDescriptors() { return _ori; }
    inline int*        getReverseMap()  { return _rev; }
    inline void        setDevice( int d ) { _device = d; }
};

} // namespace popsift
