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
void* get_pinned( size_t bytes, size_t* cap );
void  put_pinned( void* p, size_t cap );
/// bytes of pinned buffers currently handed out (not sitting in the pool)
size_t pinned_in_use( );

} // namespace pool
} // namespace popsift
