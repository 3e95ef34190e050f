// host_pool.h -- process-wide pools of result buffers (internal to libpopsift).
//
// Why: in a process with a live GPU context every munmap / mmap of a multi-megabyte result array costs
// close to a millisecond (MMU-notifier round trip), and pinned, GPU-mapped memory (hipHostMalloc) costs
// more.  FeaturesHost objects are created and deleted once per image, so their arrays are recycled:
//   plain pool   Feature arrays (72 B records with host pointers, written by the CPU)
//   pinned pool  descriptor arrays: the GPU writes the descriptors of a frame straight into the buffer
//                that the FeaturesHost of that frame will own (psx_attach_export), no host copy
#pragma once

#include <cstddef>

namespace popsift {
namespace pool {

/// a buffer of at least `bytes`; *cap receives its real size.  nullptr when out of memory.
void* get_plain( size_t bytes, size_t* cap );
void  put_plain( void* p, size_t cap );
/// Pinned buffers come from ONE POOL PER DEVICE (own mutex, own free list, pages on the NUMA node of the device's
/// PCIe root): get_pinned serves the calling thread's device (set_thread_device; -1 = the pool of no device),
/// put_pinned returns a buffer to the pool it came from, whichever thread calls it.  A free list is bounded by bytes
/// (POPSIFT_POOL_FREE_MB), so a steady stream never reaches hipHostFree / hipHostMalloc.
/// Best effort: bind the calling thread to the CPUs local to the device's PCIe root (sysfs local_cpulist, read and
/// parsed once per device; POPSIFT_NUMA_PIN=0 disables it).  false = affinity untouched.
bool  pin_thread_to_device_cpus( int device );
void  set_thread_device( int device );
int   thread_device( );
void* get_pinned( size_t bytes, size_t* cap );
void  put_pinned( void* p, size_t cap );
/// bytes of pinned buffers currently handed out by the calling thread's device pool (not sitting in the pool)
size_t pinned_in_use( );

struct Stats { long allocs = 0, frees = 0, hits = 0, free_buffers = 0; size_t free_bytes = 0, in_use = 0; };
/// counters of one device's pinned pool (device < 0: summed over all pools)
Stats pinned_stats( int device );

/// sets the calling thread's pool device for a scope (PopSift::enqueue runs on the caller's thread)
struct DeviceScope
{
    int old;
    explicit DeviceScope( int device ) : old( thread_device() ) { set_thread_device( device ); }
    ~DeviceScope() { set_thread_device( old ); }
};

} // namespace pool
} // namespace popsift
