// CPU test of the host buffer pools (popsift_amd/csrc/host/host_pool.h; internal header): the size-ordered free list
// hands out the SMALLEST adequate buffer, never one more than 4x + 4 MB too large, counts hits / allocations, and the
// NUMA helper refuses devices it knows nothing about without touching the thread's affinity (round 5, after the
// advisor's findings: multimap instead of a linear scan, reentrant cpulist parser).  Pageable pool only: no GPU needed.
#include "host_pool.h"

#include <sched.h>

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static int fails = 0;
#define CHECK(c) do { if(!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while(0)

int main()
{
    using namespace popsift::pool;
    size_t c1 = 0, c2 = 0, c3 = 0;
    void* a = get_plain( 1 << 20, &c1 );            // 1 MB -> 2 MB (25 % slack, whole megabytes)
    void* b = get_plain( 10 << 20, &c2 );           // 10 MB -> 13 MB
    void* c = get_plain( 3 << 20, &c3 );            // 3 MB -> 4 MB
    CHECK( a && b && c && c1 >= (1u << 20) && c2 >= (10u << 20) && c3 >= (3u << 20) );
    memset( a, 1, c1 ); memset( b, 2, c2 ); memset( c, 3, c3 );
    put_plain( b, c2 ); put_plain( a, c1 ); put_plain( c, c3 );
    // a 2.5 MB request: the 4 MB buffer is the smallest adequate one (not the 13 MB one, not the 2 MB one)
    size_t cc = 0;
    void* d = get_plain( (size_t)( 2.5 * ( 1 << 20 ) ), &cc );
    CHECK( d == c && cc == c3 );
    // a tiny request must not take the 13 MB buffer (more than 4x + 4 MB too large) when the 2 MB one is there
    size_t ce = 0;
    void* e = get_plain( 100, &ce );
    CHECK( e == a && ce == c1 );
    // nothing adequate and close enough is left for 100 bytes: a fresh buffer, the 13 MB one stays pooled
    size_t cf = 0;
    void* f = get_plain( 100, &cf );
    CHECK( f != b && f != nullptr );
    size_t cg = 0;
    void* g = get_plain( 9 << 20, &cg );
    CHECK( g == b && cg == c2 );
    put_plain( d, cc ); put_plain( e, ce ); put_plain( f, cf ); put_plain( g, cg );

    // many threads at once (the pools are shared by workers and callers)
    std::vector<std::thread> ts;
    for( int t = 0; t < 8; t++ )
        ts.emplace_back( []{ for( int i = 0; i < 2000; i++ ) { size_t cap = 0; void* p = get_plain( (size_t)( 1 + i % 7 ) << 18, &cap ); if( p ) { ((char*)p)[0] = 1; put_plain( p, cap ); } } } );
    for( auto& t : ts ) t.join();

    // the NUMA helper: an unknown device leaves the affinity alone and says so
    cpu_set_t before, after;
    CHECK( sched_getaffinity( 0, sizeof(before), &before ) == 0 );
    CHECK( !pin_thread_to_device_cpus( -1 ) );
    CHECK( !pin_thread_to_device_cpus( 1000 ) );
    CHECK( sched_getaffinity( 0, sizeof(after), &after ) == 0 && CPU_EQUAL( &before, &after ) );

    // DeviceScope restores the thread's pool device
    set_thread_device( 3 );
    { DeviceScope s( 5 ); CHECK( thread_device() == 5 ); }
    CHECK( thread_device() == 3 );
    std::printf( fails ? "FAILED %d\n" : "ALL OK\n", fails );
    return fails ? 1 : 0;
}
