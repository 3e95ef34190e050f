// valu_rate.hip -- issue rate of v_pk_fma_f32 / v_fma_f32 / v_pk_add_f32 on gfx950 (cycles per wave-instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float g)
{
    v2f a[8];
    for (int i = 0; i < 8; i++) a[i] = (v2f){(float)threadIdx.x + i, 1.0f + i};
    float s[16];
    for (int i = 0; i < 16; i++) s[i] = threadIdx.x * 0.5f + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], (v2f){g, g}, a[(i + 1) & 7]);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 16; i++) s[i] = fmaf(s[i], g, s[(i + 1) & 15]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = a[i] + a[(i + 3) & 7];
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    for (int i = 0; i < 16; i++) r += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)(t1 - t0) * 0.0f;
}
template <int MODE> void run(const char* name, int wg_per_cu, int ninst_per_iter)
{
    float* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, 100, 1.0001f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: wg_per_cu waves (one wave of each WG per SIMD), each iters * ninst wave-instructions
    double inst_per_simd = (double)wg_per_cu * iters * ninst_per_iter;
    printf("%-12s waves/SIMD=%d: %.3f ms -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz, %.2f @2.0GHz)\n", name, wg_per_cu, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, ms * 1e6 / inst_per_simd * 2.0);
    hipFree(d);
}
int main()
{
    for (int w = 1; w <= 4; w *= 2) { run<0>("v_pk_fma_f32", w, 32); run<1>("v_fma_f32", w, 64); run<2>("v_pk_add_f32", w, 32); }
    return 0;
}
