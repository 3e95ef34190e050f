// copy_variants.hip -- which streaming-copy shape reaches the HBM copy ceiling on MI355X?  (tools/ubench, measurement only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_gs(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(r[u], d + i + u * stride); else d[i + u * stride] = r[u]; }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
// one block copies a contiguous chunk (block-contiguous): each thread U x 16 B, consecutive in the chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_bc(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n4)
{
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    v4f r[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256 < n4) r[u] = NT ? __builtin_nontemporal_load(s + base + u * 256) : s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256 < n4) { if (NT) __builtin_nontemporal_store(r[u], d + base + u * 256); else d[base + u * 256] = r[u]; }
}
template <class F> float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    v4f *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
#define GS(U, NT, G) { float ms = timeit([&] { hipLaunchKernelGGL((k_gs<U, NT>), dim3(G), dim3(256), 0, 0, s, d, n4); }, 10); \
    printf("grid-stride U=%d NT=%d grid=%6d : %.3f ms  %.0f GB/s\n", U, NT, G, ms, 2.0 * bytes / ms / 1e6); }
#define BC(U, NT) { int G = (int)((n4 + 256 * U - 1) / (256 * U)); float ms = timeit([&] { hipLaunchKernelGGL((k_bc<U, NT>), dim3(G), dim3(256), 0, 0, s, d, n4); }, 10); \
    printf("block-contig U=%d NT=%d grid=%6d : %.3f ms  %.0f GB/s\n", U, NT, G, ms, 2.0 * bytes / ms / 1e6); }
    GS(4, false, 2048) GS(4, true, 2048) GS(4, false, 1024) GS(4, false, 4096) GS(4, false, 8192) GS(8, false, 2048) GS(2, false, 4096) GS(1, false, 8192) GS(1, false, 16384)
    GS(8, true, 4096) GS(4, true, 8192)
    BC(1, false) BC(2, false) BC(4, false) BC(8, false) BC(4, true) BC(8, true)
    { float ms = timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, 10); printf("hipMemcpyAsync D2D : %.3f ms %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6); }
    // read-only and write-only rates
    return 0;
}
