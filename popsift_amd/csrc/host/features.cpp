// features.cpp -- FeaturesHost / FeaturesDev / Feature::print
// (reference behaviour: features.cu:27-128, 306-330)
#include <cstdio>
#include "popsift/features.h"
#include "host_pool.h"
#include "popsift/sift_extremum.h"

#include "popsift_hip.h"
#include <cstddef>

#include <algorithm>
#include <cerrno>
#include <mutex>
#include <utility>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <unistd.h>

namespace popsift {

namespace {
[[noreturn]] void fatal( const char* file, int line, const std::string& msg )
{
    std::ostringstream o;
    o << file << ":" << line << std::endl << "    " << msg;
    throw std::runtime_error( o.str() );
}
void* page_alloc( size_t bytes )
{
    void* p = nullptr;
    const size_t page = (size_t)sysconf( _SC_PAGESIZE );
    const int err = posix_memalign( &p, page, bytes ? bytes : page );
    if( err != 0 ) { errno = err; return nullptr; }
    return p;
}
} // namespace

FeaturesBase::FeaturesBase( ) : _num_ext( 0 ), _num_ori( 0 ) { }
FeaturesBase::~FeaturesBase( ) = default;

// ---- buffer pools (host_pool.h) -------------------------------------------------------------------
namespace pool {
namespace {
struct Pool
{
    std::mutex                            m;
    std::vector<std::pair<void*, size_t>> free_list;
    bool                                  pinned;
    size_t                                in_use = 0;      // bytes handed out
    explicit Pool( bool p ) : pinned( p ) { }

    void* get( size_t bytes, size_t* cap )
    {
        if( bytes == 0 ) bytes = 1;
        {
            std::lock_guard<std::mutex> g( m );
            int best = -1;
            for( size_t i = 0; i < free_list.size(); i++ )
                if( free_list[i].second >= bytes && free_list[i].second <= 4 * bytes + ( 4u << 20 ) &&
                    ( best < 0 || free_list[i].second < free_list[best].second ) ) best = (int)i;
            if( best >= 0 ) {
                void* p = free_list[best].first; *cap = free_list[best].second;
                free_list.erase( free_list.begin() + best );
                in_use += *cap;
                return p;
            }
        }
        const size_t mb = (size_t)1 << 20;
        const size_t c = ( ( bytes + bytes / 4 + mb - 1 ) / mb ) * mb;       // 25 % slack, whole megabytes
        void* p = nullptr;
        if( pinned ) { if( psx_host_alloc( c, &p ) != PSX_OK ) return nullptr; }
        else         { if( posix_memalign( &p, 4096, c ) != 0 ) return nullptr; }
        *cap = c;
        { std::lock_guard<std::mutex> g( m ); in_use += c; }
        return p;
    }
    void put( void* p, size_t cap )
    {
        if( p == nullptr ) return;
        {
            std::lock_guard<std::mutex> g( m );
            in_use -= std::min( in_use, cap );
            if( free_list.size() < 32 ) { free_list.emplace_back( p, cap ); return; }
        }
        if( pinned ) psx_host_free( p ); else free( p );
    }
};
Pool& plain()  { static Pool* p = new Pool( false ); return *p; }     // never destroyed: objects may outlive main()
Pool& pinned() { static Pool* p = new Pool( true );  return *p; }
} // namespace
void* get_plain( size_t bytes, size_t* cap )  { return plain().get( bytes, cap ); }
void  put_plain( void* p, size_t cap )        { plain().put( p, cap ); }
void* get_pinned( size_t bytes, size_t* cap ) { return pinned().get( bytes, cap ); }
void  put_pinned( void* p, size_t cap )       { pinned().put( p, cap ); }
size_t pinned_in_use( )                       { std::lock_guard<std::mutex> g( pinned().m ); return pinned().in_use; }
} // namespace pool

FeaturesHost::FeaturesHost( ) : _ext( nullptr ), _ori( nullptr ), _ext_cap( 0 ), _ori_cap( 0 ) { }

FeaturesHost::FeaturesHost( int num_ext, int num_ori ) : _ext( nullptr ), _ori( nullptr ), _ext_cap( 0 ), _ori_cap( 0 )
{
    reset( num_ext, num_ori );
}

FeaturesHost::~FeaturesHost( )
{
    release();
}

void FeaturesHost::release( )
{
    if( _ext_cap ) pool::put_plain( _ext, _ext_cap ); else free( _ext );
    if( _ori_cap ) pool::put_pinned( _ori, _ori_cap ); else free( _ori );
    _ext = nullptr; _ori = nullptr; _ext_cap = _ori_cap = 0;
}

void FeaturesHost::adopt( int num_ext, int num_ori, Feature* ext, size_t ext_cap, Descriptor* ori, size_t ori_cap )
{
    release();
    _ext = ext; _ext_cap = ext_cap;
    _ori = ori; _ori_cap = ori_cap;
    setFeatureCount( num_ext );
    setDescriptorCount( num_ori );
}

void FeaturesHost::reset( int num_ext, int num_ori )
{
    release();

    _ext = (Feature*)page_alloc( (size_t)num_ext * sizeof(Feature) );
    if( _ext == nullptr ) {
        std::ostringstream ss;
        ss << "Runtime error:" << std::endl
           << "    Failed to (re)allocate memory for downloading " << num_ext << " features";
        fatal( __FILE__, __LINE__, ss.str() );
    }
    _ori = (Descriptor*)page_alloc( (size_t)num_ori * sizeof(Descriptor) );
    if( _ori == nullptr ) {
        std::ostringstream ss;
        ss << "Runtime error:" << std::endl
           << "    Failed to (re)allocate memory for downloading " << num_ori << " descriptors";
        fatal( __FILE__, __LINE__, ss.str() );
    }
    setFeatureCount( num_ext );
    setDescriptorCount( num_ori );
}

void FeaturesHost::pin( )   { }
void FeaturesHost::unpin( ) { }

void FeaturesHost::print( std::ostream& ostr, bool write_as_uchar ) const
{
    for( int i = 0; i < size(); i++ ) _ext[i].print( ostr, write_as_uchar );
}

std::ostream& operator<<( std::ostream& ostr, const FeaturesHost& feature )
{
    feature.print( ostr, false );
    return ostr;
}

// output-features.txt line format (features.cu:310-330): x y 1/s^2 0 1/s^2 d0..d127
void Feature::print( std::ostream& ostr, bool write_as_uchar ) const
{
    const float sigval = 1.0f / ( sigma * sigma );
    for( int ori = 0; ori < num_ori; ori++ ) {
        ostr << xpos << " " << ypos << " " << sigval << " 0 " << sigval << " ";
        if( write_as_uchar ) {
            for( int i = 0; i < 128; i++ ) ostr << roundf( desc[ori]->features[i] ) << " ";
        } else {
            ostr << std::setprecision(3);
            for( int i = 0; i < 128; i++ ) ostr << desc[ori]->features[i] << " ";
            ostr << std::setprecision(6);
        }
        ostr << std::endl;
    }
}

std::ostream& operator<<( std::ostream& ostr, const Feature& feature )
{
    feature.print( ostr, false );
    return ostr;
}

// FeaturesDev hands out device arrays of popsift::Feature; the C-ABI fills them as psx_feature_dev
static_assert( sizeof(Feature) == sizeof(psx_feature_dev), "popsift::Feature and psx_feature_dev must have the same layout" );
static_assert( offsetof(Feature, desc) == offsetof(psx_feature_dev, desc), "popsift::Feature::desc offset" );
static_assert( offsetof(Feature, orientation) == offsetof(psx_feature_dev, orientation), "popsift::Feature::orientation offset" );

FeaturesDev::FeaturesDev( ) : _ext( nullptr ), _ori( nullptr ), _rev( nullptr ), _device( 0 ) { }

FeaturesDev::FeaturesDev( int num_ext, int num_ori ) : _ext( nullptr ), _ori( nullptr ), _rev( nullptr ), _device( 0 )
{
    reset( num_ext, num_ori );
}

FeaturesDev::~FeaturesDev( )
{
    psx_dev_free( _device, _ext );
    psx_dev_free( _device, _ori );
    psx_dev_free( _device, _rev );
}

void FeaturesDev::reset( int num_ext, int num_ori )
{
    psx_dev_free( _device, _ext ); _ext = nullptr;
    psx_dev_free( _device, _ori ); _ori = nullptr;
    psx_dev_free( _device, _rev ); _rev = nullptr;
    void *e = nullptr, *o = nullptr, *r = nullptr;
    if( psx_dev_alloc( _device, (size_t)num_ext * sizeof(Feature), &e ) != PSX_OK ||
        psx_dev_alloc( _device, (size_t)num_ori * sizeof(Descriptor), &o ) != PSX_OK ||
        psx_dev_alloc( _device, (size_t)num_ori * sizeof(int), &r ) != PSX_OK )
        fatal( __FILE__, __LINE__, "Runtime error:\n    Failed to allocate device memory for features" );
    _ext = (Feature*)e; _ori = (Descriptor*)o; _rev = (int*)r;
    setFeatureCount( num_ext );
    setDescriptorCount( num_ori );
}

// FeaturesDev::match (features.cu:270-304): the reference computes the match matrix on the device and
// prints one line per left descriptor from a device printf (show_distance, features.cu:227-268); same
// lines here, printed by the host from the arrays psx_match returns.
void FeaturesDev::match( FeaturesDev* other )
{
    if( other == nullptr ) fatal( __FILE__, __LINE__, "FeaturesDev::match: null argument" );
    const int l_len = getDescriptorCount();
    const int r_len = other->getDescriptorCount();
    if( l_len <= 0 ) return;
    std::vector<int>   mm( 3 * (size_t)l_len );
    std::vector<float> dd( 2 * (size_t)l_len );
    std::vector<int>   l_fem( l_len ), r_fem( r_len > 0 ? r_len : 1 );
    if( psx_match( _device, (const float*)_ori, l_len, (const float*)other->_ori, r_len, mm.data(), dd.data() ) != PSX_OK ||
        psx_dev_read( _device, l_fem.data(), _rev, (size_t)l_len * sizeof(int) ) != PSX_OK ||
        ( r_len > 0 && psx_dev_read( other->_device, r_fem.data(), other->_rev, (size_t)r_len * sizeof(int) ) != PSX_OK ) )
        fatal( __FILE__, __LINE__, "FeaturesDev::match failed" );
    for( int i = 0; i < l_len; i++ )
    {
        const int m1 = mm[3*i], m2 = mm[3*i+1];
        printf( "%s feat %4d [%4d] matches feat %4d [%4d] ( 2nd feat %4d [%4d] ) dist %.3f vs %.3f\n",
                mm[3*i+2] ? "accept" : "reject",
                l_fem[i], i,
                r_len > 0 ? r_fem[m1] : 0, m1,
                r_len > 0 ? r_fem[m2] : 0, m2,
                dd[2*i], dd[2*i+1] );
    }
}

} // namespace popsift
