// write_strips.hip -- what does a write-dominated strip kernel reach on MI355X?  (tools/ubench, measurement only)
// The store pattern of k_fixed_octave (pyramid_fixed.hip): a workgroup owns a 64-column strip of NL planes (3840 x 2160 floats,
// pitch 3840) and a chunk of rows; 16 lanes write 256 contiguous bytes of a row, a wave 4 rows, NL planes per row group.
// Variants: plain / non-temporal / write-through (sc0 sc1) stores; workgroups per CU through the grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__device__ __forceinline__ void st(v4f* p, v4f v)
{
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int MODE, int NL>
__global__ __launch_bounds__(256) void k_ws(float* base, int W, int H, int pitch, size_t plane, int nstrips, int chunk_rows, float seed)
{
    const int lid = blockIdx.x, strip = lid % nstrips, chunk = lid / nstrips;
    const int x = strip * 64 + 4 * (threadIdx.x & 15);
    const int r0 = chunk * chunk_rows, r1 = min(r0 + chunk_rows, H);
    v4f v = {seed, seed + 1, seed + 2, seed + 3};
    for (int r = r0 + (threadIdx.x >> 4); r < r1; r += 16) {
#pragma unroll
        for (int l = 0; l < NL; l++) st<MODE>(reinterpret_cast<v4f*>(base + l * plane + (size_t)r * pitch + x), v);
        v.x += 1.0f;
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
    const int W = 3840, H = 2160, pitch = 3840, NLMAX = 6;
    const size_t plane = (size_t)pitch * H;
    float* d; hipMalloc(&d, plane * 4 * NLMAX * 2);
#define RUN(MODE, NL, CR) { const int nstrips = W / 64, nch = (H + CR - 1) / CR; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_ws<MODE, NL>), dim3(nstrips * nch), dim3(256), 0, 0, d, W, H, pitch, plane, nstrips, CR, 1.0f); }, 20); \
    printf("mode %d planes %d chunk_rows %4d grid %5d : %7.2f us  %.0f GB/s\n", MODE, NL, CR, nstrips * nch, ms * 1e3, (double)NL * W * H * 4 / ms / 1e6); }
    RUN(0, 6, 132) RUN(1, 6, 132) RUN(2, 6, 132)
    RUN(0, 6, 64) RUN(1, 6, 64) RUN(2, 6, 64)
    RUN(0, 6, 32) RUN(2, 6, 32)
    RUN(0, 6, 270) RUN(2, 6, 270)
    RUN(0, 1, 132) RUN(2, 1, 132) RUN(0, 1, 32) RUN(2, 1, 32)
    return 0;
}
