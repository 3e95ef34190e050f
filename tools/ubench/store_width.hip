// store_width.hip -- does the width / scope of the plane stores matter on MI355X?  (tools/ubench, measurement only)
// A streaming copy with 16-byte loads and (a) one 16-byte store, (b) two 8-byte stores per thread, each as plain and as
// system-scope (sc0 sc1, write-through) stores; plus the write-only forms.  k_blur stores 8 bytes per lane with sc0 sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool SYS> __device__ __forceinline__ void st16(v4f* p, v4f v)
{
    if (SYS) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); else *p = v;
}
template <bool SYS> __device__ __forceinline__ void st8(v2f* p, v2f v)
{
    if (SYS) { unsigned long long b; __builtin_memcpy(&b, &v, 8); __hip_atomic_store((unsigned long long*)p, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    else *p = v;
}
// MODE 0: 16-byte store, 1: two 8-byte stores (each wave instruction covers 512 contiguous bytes); LOAD: copy or write-only
template <int MODE, bool SYS, bool LOAD>
__global__ __launch_bounds__(256) void k(const v4f* __restrict__ s, float* __restrict__ d, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    v4f r = LOAD ? s[i] : (v4f){1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    if (MODE == 0) st16<SYS>((v4f*)d + i, r);
    else {
        v2f* d2 = (v2f*)d + (size_t)blockIdx.x * 512 + threadIdx.x;
        st8<SYS>(d2, (v2f){r.x, r.y});
        st8<SYS>(d2 + 256, (v2f){r.z, r.w});
    }
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
    for (int pass = 0; pass < 2; pass++) {
        const size_t bytes = pass == 0 ? ((size_t)1 << 30) : (size_t)3840 * 2160 * 4;
        const size_t n4 = bytes / 16;
        v4f* s; float* d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
        const int G = (int)((n4 + 255) / 256);
        const int reps = pass == 0 ? 10 : 200;
        printf("---- %zu MB per buffer\n", bytes >> 20);
#define RUN(MODE, SYS, LOAD, name) { float ms = timeit([&] { hipLaunchKernelGGL((k<MODE, SYS, LOAD>), dim3(G), dim3(256), 0, 0, s, d, n4); }, reps); \
        printf("%-34s : %8.2f us  %6.0f GB/s\n", name, ms * 1e3, (LOAD ? 2.0 : 1.0) * bytes / ms / 1e6); }
        RUN(0, false, true,  "copy  ld16 st16 plain")
        RUN(0, true,  true,  "copy  ld16 st16 sc0sc1")
        RUN(1, false, true,  "copy  ld16 2xst8 plain")
        RUN(1, true,  true,  "copy  ld16 2xst8 sc0sc1")
        RUN(0, false, false, "write st16 plain")
        RUN(0, true,  false, "write st16 sc0sc1")
        RUN(1, false, false, "write 2xst8 plain")
        RUN(1, true,  false, "write 2xst8 sc0sc1")
        hipFree(s); hipFree(d);
    }
    return 0;
}
