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
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <cstring>
#include <cctype>
#include <cstdint>
#include <sched.h>
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
const int MAX_POOLS = 33;                       // devices 0..31, and one pool for "no device named"

// pinned bytes a pool may keep on its free list (POPSIFT_POOL_FREE_MB, default 2048): the list is bounded by BYTES,
// sized for `jobs outstanding x (image + result)` of a replica with head-room, not by a buffer count -- a count of 32
// made a replica with 24 jobs outstanding (~40 live buffers) hipHostFree / hipHostMalloc its surplus on every
// drain / refill burst
size_t free_limit()
{
    static const size_t lim = []{ const char* e = getenv( "POPSIFT_POOL_FREE_MB" ); const long v = e ? atol( e ) : 2048; return (size_t)( v < 0 ? 0 : v ) << 20; }();
    return lim;
}

// "0-3,8,10-11\n" -> the listed CPUs that the process may run on.  No strtok: several workers, the pool's helper threads
// and the caller of enqueue() parse at the same moment during warm-up, and strtok keeps process-global state.
int parse_cpulist( const char* s, const cpu_set_t& allowed, cpu_set_t* want )
{
    CPU_ZERO( want );
    int n = 0;
    while( *s != 0 ) {
        while( *s == ',' || *s == '\n' || *s == ' ' ) s++;
        if( *s == 0 ) break;
        char* end = nullptr;
        long a = strtol( s, &end, 10 );
        if( end == s ) { while( *s != 0 && *s != ',' ) s++; continue; }     // not a number: skip the token
        long b = a;
        s = end;
        if( *s == '-' ) { b = strtol( s + 1, &end, 10 ); if( end == s + 1 ) b = a; s = end; }
        while( *s != 0 && *s != ',' ) s++;
        for( long c = a < 0 ? 0 : a; c <= b && c < CPU_SETSIZE; c++ )
            if( CPU_ISSET( (int)c, &allowed ) ) { CPU_SET( (int)c, want ); n++; }
    }
    return n;
}

// CPUs local to each device's PCIe root, read and parsed ONCE per device (sysfs does not change under us)
struct DeviceCpus { std::once_flag once; bool valid = false; cpu_set_t set; };
DeviceCpus& device_cpus( int device )
{
    static DeviceCpus* tab = new DeviceCpus[MAX_POOLS];
    DeviceCpus& d = tab[ ( device >= 0 && device < MAX_POOLS - 1 ) ? device : MAX_POOLS - 1 ];
    if( device < 0 || device >= MAX_POOLS - 1 ) return d;           // never valid
    std::call_once( d.once, [&d, device]{
        const char* e = getenv( "POPSIFT_NUMA_PIN" );
        if( e != nullptr && e[0] == '0' ) return;
        char bus[64];
        if( psx_device_pci( device, bus, sizeof(bus) ) != PSX_OK || bus[0] == 0 ) return;
        for( char* c = bus; *c; c++ ) *c = (char)tolower( *c );
        FILE* f = fopen( ( std::string( "/sys/bus/pci/devices/" ) + bus + "/local_cpulist" ).c_str(), "r" );
        if( f == nullptr ) return;
        char line[4096] = { 0 };
        const bool got = fgets( line, sizeof(line), f ) != nullptr;
        fclose( f );
        // the PROCESS's mask (its main thread's: what a launcher such as taskset / numactl / bench.py's pinning set), not the
        // calling thread's: the first caller may be a worker that is already bound to another device's socket, and its narrow
        // mask would truncate -- or empty -- this device's set for the rest of the process
        cpu_set_t allowed;
        if( !got || sched_getaffinity( getpid(), sizeof(allowed), &allowed ) != 0 ) return;
        d.valid = parse_cpulist( line, allowed, &d.set ) > 0;
    } );
    return d;
}
bool multi_node_host() { static const bool m = access( "/sys/devices/system/node/node1", F_OK ) == 0; return m; }

void forget_pinned( void* p );      // drops the buffer from the pointer -> device registry (it is about to be freed)

struct Pool
{
    std::mutex                            m;
    std::multimap<size_t, void*>          free_list;        // by capacity: the smallest adequate buffer in O(log n)
    const bool                            pinned;
    const int                             device;          // -1: none
    size_t                                in_use = 0;      // bytes handed out
    size_t                                free_bytes = 0;  // bytes on the free list
    Stats                                 st;
    Pool( bool p, int d ) : pinned( p ), device( d ) { }

    // Where the pages of a pinned buffer land: on the NUMA node the HIP runtime picks for the calling thread -- by its CPU
    // affinity or by its current device, depending on the runtime.  The caller of enqueue() runs anywhere, so on a
    // multi-socket host a pool miss allocates from a short-lived helper thread that is bound to the CPUs of the device's
    // PCIe root AND has made the device current (psx_host_alloc_near): socket-1 GPUs then DMA to socket-1 memory under
    // either rule.  Only pool misses pay for it (warm-up).
    void* alloc_pinned( size_t c )
    {
        void* p = nullptr;
        DeviceCpus& dc = device_cpus( device );
        // single-node host (or CPUs unknown): placement does not matter, and the CALLER's current device must not change
        if( !multi_node_host() || !dc.valid ) { if( psx_host_alloc( c, &p ) != PSX_OK ) return nullptr; return p; }
        const int dev = device;
        const cpu_set_t* set = &dc.set;
        std::thread t( [&p, c, dev, set]{ (void)sched_setaffinity( 0, sizeof(cpu_set_t), set ); if( psx_host_alloc_near( dev, c, &p ) != PSX_OK ) p = nullptr; } );
        t.join();
        return p;
    }

    void* get( size_t bytes, size_t* cap )
    {
        if( bytes == 0 ) bytes = 1;
        {
            std::lock_guard<std::mutex> g( m );
            auto it = free_list.lower_bound( bytes );
            if( it != free_list.end() && it->first <= 4 * bytes + ( 4u << 20 ) ) {
                void* p = it->second; *cap = it->first;
                free_list.erase( it );
                in_use += *cap; free_bytes -= *cap; st.hits++;
                return p;
            }
        }
        const size_t mb = (size_t)1 << 20;
        const size_t c = ( ( bytes + bytes / 4 + mb - 1 ) / mb ) * mb;       // 25 % slack, whole megabytes
        void* p = nullptr;
        if( pinned ) { p = alloc_pinned( c ); if( p == nullptr ) return nullptr; }
        else         { if( posix_memalign( &p, 4096, c ) != 0 ) return nullptr; }
        *cap = c;
        { std::lock_guard<std::mutex> g( m ); in_use += c; st.allocs++; }
        return p;
    }
    void put( void* p, size_t cap )
    {
        if( p == nullptr ) return;
        {
            std::lock_guard<std::mutex> g( m );
            in_use -= std::min( in_use, cap );
            if( free_bytes + cap <= free_limit() ) { free_list.emplace( cap, p ); free_bytes += cap; return; }
            st.frees++;
        }
        if( pinned ) { forget_pinned( p ); psx_host_free( p ); } else free( p );      // only beyond the byte bound: never in a steady stream
    }
    Stats stats()
    {
        std::lock_guard<std::mutex> g( m );
        Stats s = st; s.free_bytes = free_bytes; s.in_use = in_use; s.free_buffers = (long)free_list.size();
        return s;
    }
};

Pool& plain()  { static Pool* p = new Pool( false, -1 ); return *p; }     // never destroyed: objects may outlive main()
// one pinned pool per device: replicas on different devices share neither a mutex nor buffers (a buffer allocated
// next to GPU 0 is never handed to GPU 5)
Pool& pinned( int device )
{
    static Pool** pools = []{ Pool** a = new Pool*[MAX_POOLS]; for( int i = 0; i < MAX_POOLS; i++ ) a[i] = new Pool( true, i < MAX_POOLS - 1 ? i : -1 ); return a; }();
    return *pools[ ( device >= 0 && device < MAX_POOLS - 1 ) ? device : MAX_POOLS - 1 ];
}

// which pool a live pinned buffer belongs to: objects (SiftJob, FeaturesHost) outlive the thread that got the buffer and
// do not carry a device; 16 shards, each a tiny critical section
struct Registry
{
    std::mutex m[16];
    std::unordered_map<void*, int> map[16];
    static int shard( void* p ) { return (int)( ( (uintptr_t)p >> 12 ) & 15 ); }
    void set( void* p, int dev ) { const int s = shard( p ); std::lock_guard<std::mutex> g( m[s] ); map[s][p] = dev; }
    int  get( void* p )          { const int s = shard( p ); std::lock_guard<std::mutex> g( m[s] ); auto it = map[s].find( p ); return it == map[s].end() ? -1 : it->second; }
    void erase( void* p )        { const int s = shard( p ); std::lock_guard<std::mutex> g( m[s] ); map[s].erase( p ); }
};
Registry& registry() { static Registry* r = new Registry; return *r; }
void forget_pinned( void* p ) { registry().erase( p ); }
thread_local int t_device = -1;
} // namespace

bool  pin_thread_to_device_cpus( int device )
{
    DeviceCpus& dc = device_cpus( device );
    return dc.valid && sched_setaffinity( 0, sizeof(cpu_set_t), &dc.set ) == 0;
}
void  set_thread_device( int device )         { t_device = device; }
int   thread_device( )                        { return t_device; }
void* get_plain( size_t bytes, size_t* cap )  { return plain().get( bytes, cap ); }
void  put_plain( void* p, size_t cap )        { plain().put( p, cap ); }
void* get_pinned( size_t bytes, size_t* cap )
{
    const int dev = t_device;
    void* p = pinned( dev ).get( bytes, cap );
    if( p != nullptr ) registry().set( p, dev );
    return p;
}
void  put_pinned( void* p, size_t cap )       { if( p != nullptr ) pinned( registry().get( p ) ).put( p, cap ); }
size_t pinned_in_use( )
{
    // of the calling thread's device: the limit is a per-replica allowance
    return pinned( t_device ).stats().in_use;
}
Stats pinned_stats( int device )
{
    if( device >= 0 ) return pinned( device ).stats();
    Stats sum;
    for( int i = 0; i < MAX_POOLS; i++ ) {
        const Stats s = pinned( i < MAX_POOLS - 1 ? i : -1 ).stats();
        sum.allocs += s.allocs; sum.frees += s.frees; sum.hits += s.hits; sum.free_bytes += s.free_bytes; sum.in_use += s.in_use; sum.free_buffers += s.free_buffers;
    }
    return sum;
}
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
